#!/usr/bin/env python3
"""Encoder spatial self-attention at the C3 shape (B=64 frames, H=8, S=207, d_h=32): forward/backward time and
algorithmic TFLOP/s.  Run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` for MFMA utilisation."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib as L, ops
L.load()
L.set_mma_mode(os.environ.get("MMA", "bf16x3"))   # bf16x3: bf16-pipe kernels (attention_bs.h); f32: fp32-MFMA kernels
dev = torch.device("cuda:0")
B, H, S = (int(sys.argv[2]) if len(sys.argv) > 2 else 64), 8, (int(sys.argv[1]) if len(sys.argv) > 1 else 207)
D = H * 32
qk = torch.randn(B, S, 2 * D, device=dev, requires_grad=True)
v = torch.randn(B, S, D, device=dev, requires_grad=True)
kpm = torch.zeros(B, S, dtype=torch.bool, device=dev)
go = torch.randn(B, S, D, device=dev)
def fwd(): return ops.mha_self_packed(qk, v, kpm, 32 ** -0.5)[0]
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_f = timeit(fwd)
def fb():
    o = fwd(); o.backward(go); qk.grad = None; v.grad = None
t_fb = timeit(fb)
fl = 2.0 * 2 * B * H * S * S * 32
with torch.no_grad():
    t_inf = timeit(lambda: ops.mha_self_packed(qk.detach(), v.detach(), kpm, 32 ** -0.5)[0])
print(f"S={S}: inference fwd (no probability stash) {t_inf*1e3:.1f} us = {fl/t_inf/1e9:.1f} TF")
print(f"S={S}: fwd {t_f*1e3:.1f} us = {fl/t_f/1e9:.1f} TF (QK^T+PV, unpadded flops); fwd+bwd {t_fb*1e3:.1f} us = {3*fl/t_fb/1e9:.1f} TF")
