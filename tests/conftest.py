import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`-m "not gpu"` selects the host-emulator suite: independent, CPU-bound tests (the emulator interprets every MFMA),
    17 minutes in one process.  They run on four pytest-xdist workers unless the caller chose `-n` himself or set
    STCAT_TEST_SERIAL=1; the emulator library is built once, here, before the workers start.  The GPU selection is
    never distributed (one GPU, timing-sensitive tests)."""
    if (config.option.markexpr or "").strip() != "not gpu" or os.environ.get("STCAT_TEST_SERIAL"):
        return None
    if getattr(config.option, "numprocesses", None) or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    try:
        import xdist  # noqa: F401
    except ImportError:
        return None
    if not hasattr(config.option, "numprocesses"):
        return None
    try:
        from tests import backends
        backends._build_emu()
    except Exception:  # the tests themselves report a broken emulator build
        pass
    config.option.numprocesses = min(4, os.cpu_count() or 1)
    return None
