"""The C-ABI library builds for gfx950, loads, and exports every symbol include/stcat_hip.h declares
(no compute calls: no GPU here).  Also: the product refuses to run without a GPU / without the library."""
import ctypes
import os
import re

import pytest
import torch

import __graft_entry__ as entry
from stcat_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_and_symbols():
    entry.build()
    header = open(os.path.join(ROOT, "include", "stcat_hip.h")).read()
    declared = set(re.findall(r"\b(stcat_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(entry.LIB)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in stcat_hip.h but not exported"
    bound = set(L.SIGNATURES) | {"stcat_version", "stcat_last_error"}
    assert declared == bound, declared ^ bound


def test_invalid_arguments_report_errors():
    entry.build()
    lib = L._bind(ctypes.CDLL(entry.LIB))
    assert lib.stcat_layernorm_fwd(None, None, None, None, None, None, None, 4, 128, 1e-5, 0.0, 0, 0, None, None) == -1
    assert b"256" in lib.stcat_last_error()
    assert lib.stcat_linear_fwd(None, None, None, None, None, 8, 60, 16, 16, 60, 0, 0, 0, 0, None) == -1
    assert lib.stcat_debug_force_tile(32, 32) == -1


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L._lib = None
    L._backend = "hip"
    from stcat_amd import ops
    with pytest.raises(L.StcatHipError):
        ops.linear(torch.zeros(4, 64), torch.zeros(64, 64), torch.zeros(64))


@pytest.mark.gpu
def test_gpu_side_streams_run_beside_the_current_stream():
    """ops._pick_streams: the side stream (forward chains / forked decoder) and the weight-gradient stream it returns sit
    on hardware queues of their own — two 500 us spin kernels on any two of (current, side, weight-gradient) finish in
    well under 2 x 500 us.  (HIP maps streams onto 4 hardware queues; profiles/r03_hw_queues.log)"""
    import time

    import torch

    from stcat_amd import ops
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L._lib = None
    L._backend = "hip"
    lib = L.load()
    dev = torch.device("cuda:0")
    ops._PICKED.pop(dev, None)          # measure afresh, whatever an earlier test of this process picked ...
    ops._WGRAD_STREAMS.pop(dev, None)   # ... and let WgradStream take the new choice
    side, wg = ops.side_stream(dev, 0), ops.side_stream(dev, 1)
    rep = ops.PICK_REPORT[str(dev)]
    assert rep["probed"] and rep["concurrent_with_main"] >= 2, rep
    assert side is not wg
    main = torch.cuda.current_stream(dev)

    def run(streams):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for st in streams:
            assert lib.stcat_spin(500, st.cuda_stream) == 0
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3

    one = min(run([main]) for _ in range(3))
    for pair in ((main, side), (main, wg), (side, wg)):
        both = min(run(pair) for _ in range(3))
        assert both < 1.5 * one, (both, one, rep)
    assert min(run((main, side, wg)) for _ in range(3)) < 1.5 * one
    assert ops.WgradStream(torch.empty(1, device=dev)).side is wg


@pytest.mark.gpu
def test_gpu_stream_constructor_of_the_background_lane():
    """round 6: stcat_stream_create / stcat_stream_destroy — a HIP stream of the device's least / greatest priority or one
    restricted to n compute units (the lanes measured for the pipelined prefix, profiles/r06_prefix_pipeline.log); kernels
    run on them, a torch ExternalStream wraps them, bad arguments are refused"""
    from stcat_amd import ops
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L._lib = None
    L._backend = "hip"
    lib = L.load()
    dev = torch.device("cuda:0")
    for prio, cus in ((1, 0), (-1, 0), (0, 0), (0, 128)):
        out = ctypes.c_void_p()
        assert lib.stcat_stream_create(prio, cus, ctypes.byref(out)) == 0, lib.stcat_last_error()
        assert out.value
        st = torch.cuda.ExternalStream(out.value, device=dev)
        x = torch.zeros(1 << 16, device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        assert lib.stcat_spin(100, st.cuda_stream) == 0
        with torch.cuda.stream(st):
            y = ops.ew(L.EW_ADD, x, torch.ones_like(x))
        torch.cuda.current_stream(dev).wait_stream(st)
        assert float(y.sum()) == float(1 << 16)
        torch.cuda.synchronize(dev)
        assert lib.stcat_stream_destroy(out) == 0
    assert lib.stcat_stream_create(0, 0, None) != 0          # no place to return the handle
    out = ctypes.c_void_p()
    assert lib.stcat_stream_create(0, 5000, ctypes.byref(out)) != 0 and not out.value
