import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("PYTEST_XDIST_WORKER"):
        # four workers share the host's cores: each one's emulator launches and torch CPU ops take their share instead of
        # every worker spawning one thread per core (measured on 8 cores: the same test 30 s alone, 180-350 s oversubscribed)
        share = max(1, (os.cpu_count() or 4) // 4)
        os.environ.setdefault("STCAT_EMU_THREADS", str(share))
        os.environ.setdefault("OMP_NUM_THREADS", str(share))
        try:
            import torch
            torch.set_num_threads(share)
        except Exception:
            pass


# The GPU suite runs evidence first: kernels -> whole-model parity -> launch plans -> optimizer / loader /
# dispatcher -> data parallel -> the bench.py subprocess runs last.  Files not listed keep their place after the parity
# block.  (The CPU selection is distributed over xdist workers, where order does not matter.)
_GPU_ORDER = ("test_library.py", "test_ops.py", "test_map2d.py", "test_model_parity.py", "test_plans.py", "test_optim.py",
              "test_loader.py", "test_torch_ops.py", "test_dist.py", "test_dp_model.py", "test_bench_dp.py")


# the CPU selection is distributed over workers: its longest tests start first (longest-processing-time order)
_CPU_SLOW_FIRST = ("test_emu_plans_replay", "test_emu_prefix_pipeline_backbone", "test_emu_prefix_pipeline_equals", "test_emu_prefix_pipeline_two_forwards", "test_emu_train_mode_dropout", "test_emu_two_forwards", "test_emu_tiny_clip_bf16x6_planes",
                   "test_emu_nonsquare", "test_emu_train_mode_against_oracle", "test_oracle_against_model_fixture",
                   "test_emu_tiny_clip", "test_install_is_a_drop_in", "test_emu_pl_conv", "test_emu_map2d", "test_loss_and_grads")


def pytest_collection_modifyitems(config, items):
    if (config.option.markexpr or "").strip() == "not gpu":
        def slow_rank(item):
            for i, key in enumerate(_CPU_SLOW_FIRST):
                if key in item.name:
                    return i
            return len(_CPU_SLOW_FIRST)
        items.sort(key=slow_rank)
        return

    def rank(item):
        name = os.path.basename(str(item.fspath))
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else _GPU_ORDER.index("test_plans.py")
    items.sort(key=rank)          # stable: the order inside a file is the file's own


@pytest.fixture(autouse=True)
def _lean_gpu_process(request):
    """after every GPU test the pytest process hands its cached device memory back: the full-size cases (C3 / C5) leave
    > 100 GB in PyTorch's caching allocator, and later tests start CHILD processes on the same GPU (two-rank DP, bench.py)"""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        gc.collect()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
                if os.environ.get("STCAT_TEST_MEMLOG"):      # leak hunting: live / reserved device memory after every test
                    with open(os.environ["STCAT_TEST_MEMLOG"], "a") as f:
                        f.write(f"{torch.cuda.memory_allocated() / 2**30:8.2f} GiB live {torch.cuda.memory_reserved() / 2**30:8.2f} "
                                f"GiB reserved after {request.node.name}\n")
        except Exception:
            pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`-m "not gpu"` selects the host-emulator suite: independent, CPU-bound tests (the emulator interprets every MFMA),
    ~12 minutes in one process (round 4: the model-level emulator tests run a 5-bottleneck ResNet).  They run on four pytest-xdist workers unless the caller chose `-n` himself or set
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
