#!/usr/bin/env python3
"""What would the encoder's big-M Linear layers gain on the plane kernels?  

Times, at the C3 token count (M = 64 x 207 = 13248 rows), every Linear shape of a spatial encoder layer three ways:
  * today's path: `stcat_linear_fwd / _dgrad / _wgrad` on fp32 tensors (mode bf16x6p: six-product split in the kernel),
  * the plane GEMMs `stcat_pl_conv_fwd / _dgrad / _wgrad` on operands that ARE planes already (what a fused producer
    would hand over), and
  * the split pass `stcat_pl_split` that a non-fused producer would cost on top.
Operand sets are rotated past the Infinity Cache (step-like).  Usage (GPU box): python tools/bench_linear_planes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib as L, ops  # noqa: E402

M = 64 * 207
SHAPES = [("qk 256>512", 256, 512), ("v/out 256>256", 256, 256), ("ffn1 256>2048", 256, 2048), ("ffn2 2048>256", 2048, 256)]
R = 4


def timeit(fn, iters=12):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    L.load()
    L.set_mma_mode("bf16x6p")
    dev = torch.device("cuda:0")
    print(f"# M = {M} rows, mode bf16x6p, us per launch (TF algorithmic)")
    tot = {"lin": 0.0, "pl": 0.0, "split": 0.0}
    for name, K, N in SHAPES:
        flop = 2.0 * M * N * K
        w = torch.randn(N, K, device=dev) * K ** -0.5
        b = torch.randn(N, device=dev)
        xs = [torch.randn(M, K, device=dev) for _ in range(R)]
        gs = [torch.randn(M, N, device=dev) for _ in range(R)]
        it = [0]

        def nxt(lst):
            it[0] += 1
            return lst[it[0] % R]
        # ---- today's kernels
        wt = ops.LINEAR_WT.get(w)
        dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        t_f = timeit(lambda: ops.linear_fwd_raw(nxt(xs), w, b, None, False))
        dx = torch.empty(M, K, device=dev)
        t_d = timeit(lambda: L.call("stcat_linear_dgrad", nxt(gs).data_ptr(), w.data_ptr(), None, wt.data_ptr(), dx.data_ptr(),
                                    M, N, K, N, K, L.stream_of(dx)))
        t_w = timeit(lambda: L.call("stcat_linear_wgrad", nxt(gs).data_ptr(), nxt(xs).data_ptr(), dw.data_ptr(), db.data_ptr(),
                                    M, N, K, N, K, L.stream_of(dx)))
        # ---- plane kernels on plane operands (1x1 "convolution" over a [1, 1, M, K] image)
        w4 = w.view(N, 1, 1, K)
        cache = ops.WeightPlanes()
        wp, wtp = cache.refresh([w4], transposed=True)
        wp, wtp = wp[w4.data_ptr()], wtp[w4.data_ptr()]
        xps = [ops.pl_split(x.view(1, 1, M, K)) for x in xs]
        gps = [ops.pl_split(g.view(1, 1, M, N)) for g in gs]
        yf = torch.empty(1, 1, M, N, device=dev)
        t_pf = timeit(lambda: ops.pl_conv_fwd_raw(nxt(xps), wp, None, b, None, 1, 0, False, planes_out=False, f32_out=True,
                                                  out=(None, yf)))
        yp = ops.Planes.empty(yf, 1, 1, M, N)
        t_pfp = timeit(lambda: ops.pl_conv_fwd_raw(nxt(xps), wp, None, b, None, 1, 0, True, planes_out=True, f32_out=False,
                                                   out=(yp, None)))
        dxo = ops.Planes.empty(yf, 1, 1, M, K)
        t_pd = timeit(lambda: ops.pl_conv_dgrad_raw(nxt(gps), wtp, (1, 1, M, K), 1, 1, 0, out=dxo))
        t_pw = float("nan")
        if K % 128 == 0 and N % 128 == 0:
            dw4 = torch.zeros(N, 1, 1, K, device=dev)
            t_pw = timeit(lambda: ops.pl_conv_wgrad_raw(nxt(gps), nxt(xps), (N, 1, 1, K), 1, 0, out=dw4))
        t_sx = timeit(lambda: ops.pl_split(nxt(xs)))
        t_sg = timeit(lambda: ops.pl_split(nxt(gs)))
        tf = lambda t: flop / t / 1e6       # noqa: E731
        print(f"{name:16s} linear fwd {t_f:6.1f} ({tf(t_f):5.1f})  dgrad {t_d:6.1f} ({tf(t_d):5.1f})  wgrad {t_w:6.1f} ({tf(t_w):5.1f})"
              f" | planes fwd->f32 {t_pf:6.1f} ({tf(t_pf):5.1f})  fwd->planes+relu {t_pfp:6.1f}  dgrad {t_pd:6.1f} ({tf(t_pd):5.1f})"
              f"  wgrad {t_pw:6.1f} ({tf(t_pw):5.1f}) | split x {t_sx:5.1f}  split g {t_sg:5.1f}")
        tot["lin"] += t_f + t_d
        tot["pl"] += t_pf + t_pd
        tot["split"] += t_sx + t_sg
        del xs, gs, xps, gps
    print(f"# critical path (fwd + dgrad) of the four shapes: linear {tot['lin']:.0f} us, planes {tot['pl']:.0f} us, "
          f"split passes if nothing is fused {tot['split']:.0f} us  (x 6 spatial layers; qk / v / out count once each here)")


if __name__ == "__main__":
    main()
