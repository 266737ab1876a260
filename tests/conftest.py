import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The GPU suite runs evidence first: kernels -> whole-model parity -> launch plans -> optimizer / loader /
# dispatcher -> data parallel -> the bench.py subprocess runs last.  Files not listed keep their place after the parity
# block.  (The CPU selection is distributed over xdist workers, where order does not matter.)
_GPU_ORDER = ("test_library.py", "test_ops.py", "test_map2d.py", "test_model_parity.py", "test_plans.py", "test_optim.py",
              "test_loader.py", "test_torch_ops.py", "test_dist.py", "test_dp_model.py", "test_bench_dp.py")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else _GPU_ORDER.index("test_plans.py")
    items.sort(key=rank)          # stable: the order inside a file is the file's own


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
