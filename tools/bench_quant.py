#!/usr/bin/env python3
"""Where does the time of one 128x128 implicit-GEMM launch go?  Same 3x3 conv at different tile counts (M) and
reduction lengths (Cin): separates the per-K-step cost from the fixed per-launch cost."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib as L, ops
L.load(os.environ.get("STCAT_LIB_OVERRIDE", L.LIB_PATH)); L.set_mma_mode("bf16x3")
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for n in (8, 32, 64, 128):
    for cin in (64, 128, 256, 512):
        x = torch.randn(n, 32, 32, cin, device=dev); w = torch.randn(256, 3, 3, cin, device=dev) * 0.02
        sc, bi = torch.ones(256, device=dev), torch.zeros(256, device=dev)
        t = timeit(lambda: ops.conv_fwd_raw(x, w, sc, bi, None, 1, 1, True))
        M = n * 1024; tiles = M // 128 * 2; steps = 9 * cin // 32
        print(f"tiles={tiles:5d} K-steps={steps:4d}: {t*1e3:7.1f} us  {2.0*M*256*9*cin/t/1e9:6.1f} TF  "
              f"{t*1e3*2.1e3/steps/max(1.0, tiles/512):7.0f} cycles per K-step and round")
