"""ctypes binding of libstcat_hip.so (C ABI declared in include/stcat_hip.h).

The product path has exactly one backend: the HIP library built for gfx950.
If it is missing, loading fails loudly — there is no CPU or PyTorch fallback.
(The CPU test-suite binds the host SIMT emulator build of the *same* kernel sources
from its own side — tests/backends.py::use_emu — nothing in the package does.)
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstcat_hip.so")

# name -> argument kinds: p = device pointer, i = int, l = long, f = float, s = stream (void*)
SIGNATURES: Dict[str, str] = {
    "stcat_frozen_bn_fold": "ppppppifs",
    "stcat_stem_fwd": "pppppiiis",
    "stcat_stem_u8_fwd": "pppppppiiis",
    "stcat_maxpool3x3s2": "ppiiiis",
    "stcat_conv_fwd": "ppppppiiiiiiiiiis",
    "stcat_conv_dgrad": "pppppppppiiiiiiiiis",
    "stcat_weight_transpose": "ppiiis",
    "stcat_weight_transpose_multi": "piis",
    "stcat_weight_transpose_entry_bytes": "",
    "stcat_conv_wgrad": "pppiiiiiiiiis",
    "stcat_act_bwd": "ppppplii" + "s",
    "stcat_pos_sine_2d": "pppiiis",
    "stcat_sine_embed_fwd": "pppis",
    "stcat_sine_embed_bwd": "ppppis",
    "stcat_linear_fwd": "pppppiiiiiiiils",
    "stcat_linear_dgrad": "pppppiiiiis",
    "stcat_linear_fwd_acc": "pppppiiiiiis",
    "stcat_linear_dgrad_acc": "ppppiiiiis",
    "stcat_linear_fwd_multi": "i" + "p" * 32 + "iii" + "s",
    "stcat_linear_dgrad_multi": "i" + "p" * 32 + "iii" + "s",
    "stcat_linear_wgrad_multi": "i" + "p" * 32 + "iii" + "s",
    "stcat_linear_fwd_drop": "ppppp" + "iiiiiii" + "fllp" + "s",
    "stcat_linear_dgrad_mask": "ppppp" + "f" + "p" + "iiiii" + "s",
    "stcat_linear_wgrad": "ppppiiiiis",
    "stcat_small_linear_fwd": "ppppiiis",
    "stcat_small_linear_bwd": "ppppppiiis",
    "stcat_colsum": "pppiis",
    "stcat_layernorm_fwd": "pppppppiif" + "fllps",
    "stcat_layernorm_bwd": "ppppppppppii" + "fllps",
    "stcat_ew": "ipppp" + "llffs",
    "stcat_ew2d": "iplplpllif" + "fs",
    "stcat_stg_loss_fwd": "p" * 13 + "f" + "iiiii" + "ppp" + "s",
    "stcat_stg_loss_bwd": "p" * 13 + "f" + "iiiii" + "ppp" + "pppp" + "s",
    "stcat_dropout": "ppplfllps",
    "stcat_mha_self_fwd": "ppppppiiiiiiif" + "fllps",
    "stcat_mha_self_bwd": "pppppppppppp" + "iiiiiiiiif" + "fllps",
    "stcat_mha_self_fwd_lse": "pppppp" + "iiiiiiif" + "fllps",
    "stcat_mha_self_bwd_lse": "pppppppppp" + "iiiiiiiiif" + "fllps",
    "stcat_mha_bs_fwd": "pppppp" + "iiiiiiif" + "fllps",
    "stcat_mha_bs_bwd": "pppppppppp" + "iiiiiiiiif" + "fllps",
    "stcat_attn_weights_mean": "ppiii" + "fllps",
    "stcat_attn_q1_fwd": "pppppppp" + "iiiiiif" + "fllps",
    "stcat_attn_q1_bwd": "pppppppppppp" + "iiiiiif" + "fllps",
    "stcat_map2d_pool": "ppiiiis",
    "stcat_map2d_cells": "pppipiiis",
    "stcat_map2d_cells_bwd": "pppippiiis",
    "stcat_map2d_pool_bwd": "pppiiiis",
    "stcat_rowscale": "pplii" + "s",
    "stcat_grad_sqnorm": "pppiips",
    "stcat_adamw_ema_step": "pppiipPPifffiffs",
    "stcat_grad_clip_scale": "pppiipfs",
    "stcat_ema_update": "pppiifs",
    "stcat_optim_table_entry_bytes": "",
    "stcat_temporal_map_argmax": "pppiis",
    "stcat_pl_conv_fwd": "pppppppppppp" + "iiiiiiiiii" + "s",
    "stcat_pl_conv_dgrad": "ppppppppppppppp" + "iiiiiiiii" + "s",
    "stcat_pl_linear_fwd": "ppppp" + "p" + "ppp" + "p" + "iiii" + "fllp" + "s",
    "stcat_pl_linear_dgrad_mask": "pppp" + "pp" + "ppp" + "iii" + "s",
    "stcat_pl_colsum": "pppiis",
    "stcat_pl_split_sum": "ppppp" + "l" + "s",
    "stcat_pl_conv_dgrad_cadd": "pppppp" + "i" + "pppp" + "iiiii" + "s",
    "stcat_pl_conv_wgrad": "pppppp" + "iiiiiiiii" + "s",
    "stcat_pl_conv_wgrad_ws": "pppppp" + "iiiiiiiii" + "pl" + "s",
    "stcat_pl_maxpool3x3s2": "pppiiiis",
    "stcat_pl_split": "pppls",
    "stcat_pl_join": "pppls",
    "stcat_pl_act_bwd": "ppppppp" + "lii" + "s",
    "stcat_pl_scale": "ppppplis",
    "stcat_weight_planes_entry_bytes": "",
    "stcat_weight_planes_multi": "piis",
    "stcat_debug_force_pl_tile": "i",
    "stcat_debug_pl_flags": "i",
    "stcat_debug_force_tile": "ii",
    "stcat_debug_streamk": "i",
    "stcat_spin": "is",
    "stcat_stream_create": "iiP",
    "stcat_stream_destroy": "P",
    "stcat_set_mma_mode": "i",
    "stcat_get_mma_mode": "",
    "stcat_set_f16_scales": "ii",
    "stcat_get_f16_scale": "i",
    # launch plans (csrc/launch_plan.h): P = host pointer, u = unsigned 64-bit, S = C string
    "stcat_plan_fn_index": "S",
    "stcat_plan_fn_nargs": "i",
    "stcat_plan_create": "",
    "stcat_plan_destroy": "P",
    "stcat_plan_add_call": "PiPiii",
    "stcat_plan_add_wait": "Pii",
    "stcat_plan_add_memset": "Ppuii",
    "stcat_plan_set_word": "Piu",
    "stcat_plan_add_yield": "Pi",
    "stcat_plan_add_reloc": "Piiu",
    "stcat_plan_size": "PPPP",
    "stcat_plan_run": "PPiPiiPP",
}
_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long, "f": ctypes.c_float, "s": ctypes.c_void_p,
       "P": ctypes.c_void_p,  # P = HOST pointer (small by-value arrays, plan handles)
       "u": ctypes.c_ulonglong, "S": ctypes.c_char_p}

EW_ADD, EW_MUL, EW_SIGMOID, EW_TANH, EW_RELU, EW_INVSIG = 0, 1, 2, 3, 4, 5
EW_SIGMOID_BWD, EW_TANH_BWD, EW_INVSIG_BWD, EW_ADD3, EW_AXPBY, EW_COPY = 6, 7, 8, 9, 10, 11

_lib: Optional[ctypes.CDLL] = None
_backend = "hip"


class StcatHipError(RuntimeError):
    pass


def _bind(lib: ctypes.CDLL) -> ctypes.CDLL:
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.argtypes = [_CT[c] for c in sig]
        fn.restype = ctypes.c_int
    lib.stcat_plan_create.restype = ctypes.c_void_p
    lib.stcat_version.restype = ctypes.c_int
    lib.stcat_last_error.restype = ctypes.c_char_p
    return lib


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise StcatHipError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). stcat_amd has no CPU/PyTorch fallback.")
        _lib = _bind(ctypes.CDLL(path))
        flags = os.environ.get("STCAT_PL_FLAGS")       # timing experiments only (stcat_debug_pl_flags: A/B of kernel variants)
        if flags:
            _lib.stcat_debug_pl_flags(int(flags))
    return _lib


def backend() -> str:
    return _backend


MMA_MODES = {"f32": 0, "bf16x3": 2, "bf16x6": 3, "bf16x3p": 4, "bf16x6p": 5, "f16x3p": 6}
PLANE_MODES = {"bf16x3p": 2, "bf16x6p": 3, "f16x3p": 2}   # mode -> 16-bit planes per backbone tensor
_mode_cache = None


def set_mma_mode(mode: str) -> None:
    """Arithmetic of the conv / Linear GEMM family: 'f32' (exact fp32 MFMA), 'bf16x3' or 'bf16x6'
    (fp32 operands split into bf16 pieces on the bf16 matrix pipe, fp32 accumulate), or 'bf16x3p': the bf16x3
    arithmetic with the backbone's activations / gradients / weights kept PRE-SPLIT as bf16 hi/lo planes in HBM
    (csrc/igemm_pl.h: LDS-DMA staged 256-wide tiles); all other GEMMs run as 'bf16x3'.  'bf16x6p' is the fp32-class
    form of that layout: THREE bf16 planes per tensor (hi + mid + lo = the fp32 value exactly), six cross terms per
    product; all other GEMMs run as 'bf16x6' and attention on the fp32 matrix pipe.  'f16x3p' (round 4, experimental): TWO
    fp16 planes per backbone tensor (22 significand bits, three products), weights / gradients scaled by powers of two
    into fp16's range (`f16_grad_scale()`); everything outside the backbone as in 'bf16x6p'."""
    global _mode_cache
    call("stcat_set_mma_mode", MMA_MODES[mode])
    _mode_cache = mode


def get_mma_mode() -> str:
    global _mode_cache
    if _mode_cache is None:
        code = load().stcat_get_mma_mode()
        _mode_cache = {v: k for k, v in MMA_MODES.items()}[code]
    return _mode_cache


def f16_grad_scale() -> float:
    """factor carried by every GRADIENT plane in mode f16x3p (1.0 in every other mode): tools / op tests that build
    gradient planes themselves multiply by it; the product path scales where gradients enter the backbone (pl_act_bwd)"""
    return float(2 ** load().stcat_get_f16_scale(1)) if get_mma_mode() == "f16x3p" else 1.0


def plane_count() -> int:
    """bf16 planes per backbone tensor in the current mode (0: the mode keeps fp32 tensors)"""
    return PLANE_MODES.get(get_mma_mode(), 0)


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def stream_of(t: torch.Tensor):
    """raw hipStream_t of torch's CURRENT stream on t's device (honours torch.cuda.stream(...) contexts)"""
    if _backend == "emu":
        return None
    # torch.cuda.current_stream() builds a Stream object (~5 us, ~1750 launches per step); the raw getter is ~0.3 us
    return torch._C._cuda_getCurrentRawStream(t.device.index if t.device.index is not None else torch.cuda.current_device())


def check_tensor(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    if _backend == "hip" and not t.is_cuda:
        raise StcatHipError(f"{name} must live on the GPU (got {t.device}); stcat_amd has no CPU path")
    if _backend == "emu" and t.is_cuda:
        raise StcatHipError("emulator backend takes CPU tensors")
    return t


RECORDER = None  # stcat_amd.plans.Recorder while a launch plan is being recorded (the calls still execute)


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.stcat_last_error()
        raise StcatHipError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")
    if RECORDER is not None:
        RECORDER.add_call(name, args)


def pin_host_threads_to_gpu(device_index: int = 0) -> Optional[str]:
    """Restrict this process to the CPUs of the NUMA node the GPU hangs off (`/sys/bus/pci/devices/<bdf>/local_cpulist`).
    The 8-GPU MI355X hosts are two-socket machines with four GPUs per socket; one process per GPU keeps its launching
    threads (python, the autograd engine thread, the HIP runtime's) on the GPU's socket, so eight ranks do not migrate
    across each other's cores.  On an otherwise idle host the step time is the same from either socket (measured:
    62.9 ms pinned, unpinned and from the remote socket) — this is hygiene for the 8-rank launch, not a speed-up.
    Returns the cpulist used, or None when the topology is not visible (then nothing changes)."""
    import torch
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            cpulist = f.read().strip()
        cpus = set()
        for part in cpulist.split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        for tid in os.listdir("/proc/self/task"):      # every thread that already exists (HIP runtime, torch pools);
            try:                                        # threads created later inherit the mask
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass
        return cpulist
    except (OSError, AttributeError, ValueError, RuntimeError):
        return None
