"""Backend selection for the kernel tests.

`hip`  : the product library stcat_amd/lib/libstcat_hip.so on cuda:0 (tests marked gpu).
`emu`  : the SAME kernel sources compiled for the host SIMT emulator (tests/emu/), CPU
         tensors — checks index logic and the autograd wiring without a GPU.
"""
import os
import subprocess

import pytest
import torch

from stcat_amd import _lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_SO = os.path.join(HERE, "emu", "_build", "libstcat_emu.so")
_emu_state = {"built": False, "error": None}


def _build_emu():
    if _emu_state["built"] or _emu_state["error"]:
        return
    srcs = [os.path.join(HERE, "emu", f) for f in ("hip_emu.h", "emu_main.cpp")]
    csrc = os.path.join(HERE, "..", "stcat_amd", "csrc")
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc)]
    srcs.append(os.path.join(HERE, "..", "include", "stcat_hip.h"))
    fresh = os.path.exists(EMU_SO) and all(os.path.getmtime(EMU_SO) >= os.path.getmtime(s) for s in srcs)
    if not fresh:
        r = subprocess.run(["sh", os.path.join(HERE, "emu", "build_emu.sh")], capture_output=True, text=True)
        if r.returncode != 0:
            _emu_state["error"] = r.stderr[-2000:]
            return
    _emu_state["built"] = True


def use_emu():
    _build_emu()
    if _emu_state["error"]:
        pytest.skip("host emulator build failed: " + _emu_state["error"])
    # bind the emulator build in place of libstcat_hip.so (test-side only: the package has no such switch)
    import ctypes
    L._lib = L._bind(ctypes.CDLL(EMU_SO))
    L._backend = "emu"
    L._mode_cache = None
    return torch.device("cpu")


def use_hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    L._lib = None
    L._backend = "hip"
    L._mode_cache = None
    L.load()
    return torch.device("cuda:0")


def both(fn):
    """Register fn(dev, big) twice: test_emu_<name> (CPU, emulator) and test_gpu_<name> (marked gpu)."""
    import sys
    mod = sys.modules[fn.__module__]
    name = fn.__name__.lstrip("_")

    def emu_test():
        fn(use_emu(), False)

    def gpu_test():
        fn(use_hip(), True)

    emu_test.__name__ = f"test_emu_{name}"
    gpu_test.__name__ = f"test_gpu_{name}"
    setattr(mod, emu_test.__name__, emu_test)
    setattr(mod, gpu_test.__name__, pytest.mark.gpu(gpu_test))
    return fn


def close(a, b, tol, what, absolute=False):
    """max |a - b| <= tol * max(1, |b|max); absolute=True drops the scale factor (the north-star bar for the model's
    box / logit tensors is an ABSOLUTE 1e-3, BASELINE.json)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = 1.0 if absolute else max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} (scale {scale:.3g}) > tol {tol}"
    return err
