#!/usr/bin/env python3
"""GPU box: device time of the stem kernel fed with fp32 NCHW frames vs. uint8 HWC frames (C3: 64 x 448 x 448)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib as L, ops  # noqa: E402

L.load()
dev = torch.device("cuda:0")
T, res = 64, 448
u8 = torch.randint(0, 256, (T, res, res, 3), dtype=torch.uint8, device=dev)
f32 = torch.randn(T, 3, res, res, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
s, b = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for mode in ("f32", "bf16x3p"):
    L.set_mma_mode(mode)
    print(mode, "stem fp32 NCHW %.3f ms | stem uint8 HWC %.3f ms" % (t(lambda: ops.stem_fwd_raw(f32, w, s, b)),
                                                                    t(lambda: ops.stem_u8_fwd_raw(u8, w, s, b))))
h = torch.randint(0, 256, (T, res, res, 3), dtype=torch.uint8).pin_memory()
hf = torch.empty(T, 3, res, res).pin_memory()
print("H2D pinned: uint8 38.5 MB %.3f ms | fp32 154 MB %.3f ms" % (t(lambda: h.to(dev, non_blocking=True)), t(lambda: hf.to(dev, non_blocking=True))))
