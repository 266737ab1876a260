"""`torch.ops.stcat_hip.*`: the C ABI of libstcat_hip.so registered with the PyTorch dispatcher (SURVEY.md §8b
"what the native library must export").

Each op is a thin schema'd entry over the same `extern "C"` call the autograd functions of `stcat_amd.ops` make
(ctypes stays the FFI underneath, include/stcat_hip.h is the contract); tensors are borrowed, outputs are allocated by
the op on the inputs' device and the launch goes to torch's current stream.  Registered for the CUDA key (the HIP
device) and — so the CPU test-suite can drive the very same entries through the host emulator build — for CPU.
Forward and backward kernels are separate ops, as §8b lists them; the autograd wiring that composes them lives in
`stcat_amd.ops`.  Importing this module registers the namespace once.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from . import ops

_LIB = torch.library.Library("stcat_hip", "DEF")
_REGISTERED = []


def _register(schema: str):
    name = schema.split("(")[0]

    def deco(fn):
        _LIB.define(schema)
        for key in ("CUDA", "CPU"):
            _LIB.impl(name, fn, key)
        _REGISTERED.append(name)
        return fn
    return deco


# ---- backbone -----------------------------------------------------------------------------------------------------
@_register("conv_bn_act_fwd(Tensor x, Tensor w_ohwi, Tensor? scale, Tensor? bias, Tensor? res, int stride, int pad, bool relu) -> Tensor")
def conv_bn_act_fwd(x, w_ohwi, scale, bias, res, stride, pad, relu):
    """NHWC conv + FrozenBN scale/bias + residual + ReLU (backbone.py:56-66, 115-119) -> NHWC"""
    return ops.conv_fwd_raw(x, w_ohwi, scale, bias, res, stride, pad, relu)


@_register("conv_dgrad(Tensor g, Tensor w_ohwi, int[] in_shape, int stride, int pad, Tensor? add, Tensor? mask_y, Tensor? mask_scale) -> Tensor")
def conv_dgrad(g, w_ohwi, in_shape, stride, pad, add, mask_y, mask_scale):
    return ops.conv_dgrad_raw(g, w_ohwi, tuple(in_shape), stride, pad, add=add, mask_y=mask_y, mask_scale=mask_scale)


@_register("conv_wgrad(Tensor g, Tensor x, int[] w_shape_ohwi, int stride, int pad) -> Tensor")
def conv_wgrad(g, x, w_shape_ohwi, stride, pad):
    return ops.conv_wgrad_raw(g, x, tuple(w_shape_ohwi), stride, pad).clone()  # (the raw op may hand out arena memory)


@_register("stem_fwd(Tensor frames_nchw, Tensor w_oihw, Tensor scale, Tensor bias) -> Tensor")
def stem_fwd(frames, w, scale, bias):
    return ops.stem_fwd_raw(frames, w, scale, bias)


@_register("maxpool3x3s2_fwd(Tensor x) -> Tensor")
def maxpool3x3s2_fwd(x):
    return ops.maxpool_raw(x)


@_register("pos_sine_2d(Tensor mask) -> Tensor")
def pos_sine_2d(mask):
    return ops.pos_sine_2d(mask)


# ---- plane-format backbone (mma mode bf16x3p): a tensor is a [2, ...] bf16 pair (hi, lo) ---------------------------
@_register("planes_split(Tensor x) -> Tensor")
def planes_split(x):
    return ops.pl_split(x).t


@_register("planes_join(Tensor planes) -> Tensor")
def planes_join(planes):
    return ops.pl_join(ops.Planes(planes))


@_register("conv_bn_act_fwd_planes(Tensor x, Tensor w, Tensor? scale, Tensor? bias, Tensor? res, int stride, int pad, bool relu) -> Tensor")
def conv_bn_act_fwd_planes(x, w, scale, bias, res, stride, pad, relu):
    y, _ = ops.pl_conv_fwd_raw(ops.Planes(x), ops.Planes(w), scale, bias, ops.Planes(res) if res is not None else None,
                               stride, pad, relu)
    return y.t


# ---- Linear / LayerNorm ---------------------------------------------------------------------------------------------
@_register("linear_bias_act_fwd(Tensor x2d, Tensor w, Tensor? bias, Tensor? res2d, bool relu) -> Tensor")
def linear_bias_act_fwd(x2d, w, bias, res2d, relu):
    return ops.linear_fwd_raw(x2d, w, bias, res2d, relu)


@_register("linear_bwd(Tensor g2d, Tensor x2d, Tensor w, bool want_bias) -> (Tensor, Tensor, Tensor)")
def linear_bwd(g, x2, w, want_bias):
    """(dx, dw, db) of y = x w^T + b (db is empty when not wanted)"""
    N, K = w.shape
    M = g.shape[0]
    st = L.stream_of(g)
    dx = torch.empty(M, K, device=g.device, dtype=torch.float32)
    dw = torch.zeros(N, K, device=g.device, dtype=torch.float32)
    db = torch.zeros(N if want_bias else 0, device=g.device, dtype=torch.float32)
    if N % 64 == 0:
        L.call("stcat_linear_dgrad", g.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), M, N, K, N, K, st)
        L.call("stcat_linear_wgrad", g.data_ptr(), x2.data_ptr(), dw.data_ptr(), L._ptr(db if want_bias else None), M, N, K,
               N, K, st)
    else:
        L.call("stcat_small_linear_bwd", g.data_ptr(), x2.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(),
               L._ptr(db if want_bias else None), M, N, K, st)
    return dx, dw, db


@_register("layernorm_residual_fwd(Tensor x, Tensor? res, Tensor gamma, Tensor beta, float eps) -> (Tensor, Tensor, Tensor)")
def layernorm_residual_fwd(x, res, gamma, beta, eps):
    D = x.shape[-1]
    x2 = x.reshape(-1, D).contiguous()
    r2 = res.reshape(-1, D).contiguous() if res is not None else None
    M = x2.shape[0]
    y = torch.empty_like(x2)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    L.call("stcat_layernorm_fwd", x2.data_ptr(), L._ptr(r2), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
           mean.data_ptr(), rstd.data_ptr(), M, D, eps, 0.0, 0, 0, None, L.stream_of(x2))
    return y.view(x.shape), mean, rstd


@_register("layernorm_residual_bwd(Tensor dy, Tensor x, Tensor? res, Tensor gamma, Tensor mean, Tensor rstd) -> (Tensor, Tensor, Tensor)")
def layernorm_residual_bwd(dy, x, res, gamma, mean, rstd):
    D = x.shape[-1]
    x2, g2 = x.reshape(-1, D).contiguous(), dy.reshape(-1, D).contiguous()
    r2 = res.reshape(-1, D).contiguous() if res is not None else None
    M = x2.shape[0]
    dz = torch.empty_like(x2)
    dgam = torch.zeros(D, device=x.device, dtype=torch.float32)
    dbet = torch.zeros(D, device=x.device, dtype=torch.float32)
    L.call("stcat_layernorm_bwd", g2.data_ptr(), x2.data_ptr(), L._ptr(r2), gamma.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), dz.data_ptr(), None, dgam.data_ptr(), dbet.data_ptr(), M, D, 0.0, 0, 0, None, L.stream_of(x2))
    return dz.view(x.shape), dgam, dbet


# ---- attention ------------------------------------------------------------------------------------------------------
@_register("mha_self_fwd(Tensor q, Tensor k, Tensor v, Tensor? key_padding_mask, float scale, bool need_weights) -> (Tensor, Tensor)")
def mha_self_fwd(q, k, v, kpm, scale, need_weights):
    """softmax(scale q k^T + key padding) v per (batch, head), head dim 32; (out, head-mean weights | empty)"""
    with torch.no_grad():
        o, w = ops.mha_self(q, k, v, kpm, scale, need_weights=need_weights)
    return o, (w if w is not None else o.new_empty(0))


@_register("mha_q1_cross_fwd(Tensor q1, Tensor? q2, Tensor k1, Tensor? k2, Tensor v, Tensor? key_padding_mask, float scale) -> Tensor")
def mha_q1_cross_fwd(q1, q2, k1, k2, v, kpm, scale):
    """time-aligned cross-attention, ONE query per batch row (query_decoder.py:386-417 / attention.py:184-393)"""
    with torch.no_grad():
        return ops.attn_q1(q1, q2, k1, k2, v, kpm, scale)


@_register("sine_embed_anchor(Tensor anchor) -> Tensor")
def sine_embed_anchor(anchor):
    with torch.no_grad():
        return ops.sine_embed(anchor)


@_register("temporal_map_argmax(Tensor pred_sted, int[] durations) -> Tensor")
def temporal_map_argmax(pred_sted, durations):
    return ops.temporal_map_argmax(pred_sted, list(durations))


def registered_ops() -> List[str]:
    return list(_REGISTERED)
