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
