#!/usr/bin/env python3
"""Accuracy of the conv GEMM kernels per mode against an fp64 convolution of the SAME (16-bit-rounded) operands:
what remains is the dropped lo*lo term and fp32 accumulation — expected ~1e-6 relative."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
L.load()


def r16(t):  # hi + lo
    h = t.to(torch.bfloat16).float()
    return h + (t - h).to(torch.bfloat16).float()


def rel(a, b):
    return ((a.double().cpu() - b).norm() / b.norm()).item()


for (n, H, W, Cin, Cout, k, stride, pad) in [(8, 14, 14, 512, 512, 3, 1, 1), (8, 14, 14, 2048, 512, 1, 1, 0),
                                              (8, 28, 28, 256, 256, 3, 1, 1), (8, 28, 28, 512, 256, 3, 2, 1)]:
    torch.manual_seed(0)
    x = r16(torch.randn(n, H, W, Cin, device=dev))
    w = r16(torch.randn(Cout, k, k, Cin, device=dev) * (Cin * k * k) ** -0.5)
    x64, w64 = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True), w.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    y64 = F.conv2d(x64, w64, stride=stride, padding=pad)
    g = r16(torch.randn(*y64.permute(0, 2, 3, 1).shape, device=dev))
    y64.backward(g.double().cpu().permute(0, 3, 1, 2))
    ref_y, ref_dx, ref_dw = y64.detach().permute(0, 2, 3, 1), x64.grad.permute(0, 2, 3, 1), w64.grad.permute(0, 2, 3, 1)
    out = []
    for mode in ("f32", "bf16x3", "bf16x3p"):
        L.set_mma_mode(mode)
        if mode == "bf16x3p":
            xp, gp = ops.pl_split(x), ops.pl_split(g)
            wp, wt = ops.WeightPlanes().refresh([w], transposed=True)
            wp, wt = wp[w.data_ptr()], wt[w.data_ptr()]
            _, y = ops.pl_conv_fwd_raw(xp, wp, None, None, None, stride, pad, False, planes_out=False, f32_out=True)
            dx = ops.pl_join(ops.pl_conv_dgrad_raw(gp, wt, x.shape, k, stride, pad))
            dw = ops.pl_conv_wgrad_raw(gp, xp, w.shape, stride, pad) if Cin % 128 == 0 else None
        else:
            y = ops.conv_fwd_raw(x, w, None, None, None, stride, pad, False)
            dx = ops.conv_dgrad_raw(g, w, x.shape, stride, pad)
            dw = ops.conv_wgrad_raw(g, x, w.shape, stride, pad)
        out.append(f"{mode}: fwd {rel(y, ref_y):.2e} dgrad {rel(dx, ref_dx):.2e} wgrad {rel(dw, ref_dw):.2e}")
    print(f"{k}x{k}/{stride} {Cin}->{Cout} {H}x{W}: " + " | ".join(out))
L.set_mma_mode("f32")
