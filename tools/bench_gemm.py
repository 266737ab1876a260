#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM family on the conv / Linear shapes of config C3 (T=64, 448x448).
Usage (GPU box):  python tools/bench_gemm.py [--mma f32|bf16x3|bf16x6] [--tile BMxBN] [--variant N]
Prints algorithmic TFLOP/s per shape for fwd / dgrad / wgrad."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib as L, ops  # noqa: E402

# (name, n, H, W, Cin, Cout, k, stride, pad)
SHAPES = [
    ("l1.conv2 3x3 64", 64, 112, 112, 64, 64, 3, 1, 1),
    ("l1.conv3 1x1 64>256", 64, 112, 112, 64, 256, 1, 1, 0),
    ("l2.conv2 3x3 128", 64, 56, 56, 128, 128, 3, 1, 1),
    ("l2.conv3 1x1 128>512", 64, 56, 56, 128, 512, 1, 1, 0),
    ("l2.conv1 1x1 512>128", 64, 56, 56, 512, 128, 1, 1, 0),
    ("l3.conv2 3x3 256", 64, 28, 28, 256, 256, 3, 1, 1),
    ("l3.conv3 1x1 256>1024", 64, 28, 28, 256, 1024, 1, 1, 0),
    ("l3.conv1 1x1 1024>256", 64, 28, 28, 1024, 256, 1, 1, 0),
    ("l4.conv2 3x3 512", 64, 14, 14, 512, 512, 3, 1, 1),
    ("l4.conv3 1x1 512>2048", 64, 14, 14, 512, 2048, 1, 1, 0),
    ("l4.conv1 1x1 2048>512", 64, 14, 14, 2048, 512, 1, 1, 0),
    ("l3.0.conv2 3x3/2 256", 64, 56, 56, 256, 256, 3, 2, 1),
    ("enc.ffn1 256>2048", 1, 1, 13248, 256, 2048, 1, 1, 0),
    ("enc.qk 256>512", 1, 1, 13248, 256, 512, 1, 1, 0),
    ("enc.out 256>256", 1, 1, 13248, 256, 256, 1, 1, 0),
    ("enc.ffn2 2048>256", 1, 1, 13248, 2048, 256, 1, 1, 0),
    ("dec.kv 256>1536", 1, 1, 13248, 256, 1536, 1, 1, 0),
]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mma", default="f32")
    ap.add_argument("--tile", default="")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--only", default="")
    ap.add_argument("--pl-tile", type=int, default=-1, help="force the plane-GEMM tile index (bf16x3p)")
    ap.add_argument("--pl-flags", type=int, default=0, help="stcat_debug_pl_flags (timing experiments)")
    ap.add_argument("--step-like", action="store_true", help="bf16x3p: operands as in the training step — the forward "
                    "reads a residual, the data gradient an `add` operand, and launches rotate over operand sets larger "
                    "than the 256 MB Infinity Cache (a step never finds its operands cached)")
    args = ap.parse_args()
    L.load(os.environ.get("STCAT_LIB_OVERRIDE", L.LIB_PATH))  # experiment builds only
    L.set_mma_mode(args.mma)
    if args.tile:
        bm, bn = [int(v) for v in args.tile.split("x")]
        L.call("stcat_debug_force_tile", bm, bn)
    L.call("stcat_debug_pl_flags", args.pl_flags)
    dev = torch.device("cuda:0")
    print(f"# mma={args.mma} pl_tile={args.pl_tile} pl_flags={args.pl_flags} tile={args.tile or 'auto'} variant={args.variant}")
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    for name, n, H, W, Cin, Cout, k, stride, pad in SHAPES:
        if args.only and args.only not in name:
            continue
        if args.mma in ("bf16x3p", "bf16x6p"):
            if n == 1:
                continue  # the Linear shapes stay on the fp32-tensor kernels
            if args.pl_tile >= 0:
                L.call("stcat_debug_force_pl_tile", args.pl_tile)
            x = torch.randn(n, H, W, Cin, device=dev)
            w = torch.randn(Cout, k, k, Cin, device=dev) * (Cin * k * k) ** -0.5
            sc, bi = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
            xp = ops.pl_split(x)
            cache = ops.WeightPlanes()
            wp, wt = cache.refresh([w], transposed=True)
            wp, wt = wp[w.data_ptr()], wt[w.data_ptr()]
            yp, _ = ops.pl_conv_fwd_raw(xp, wp, sc, bi, None, stride, pad, True)
            flop = 2.0 * yp.numel() * k * k * Cin
            R = 3 if args.step_like else 1
            sets = []
            for r in range(R):
                st = {"xp": ops.pl_split(torch.randn(n, H, W, Cin, device=dev)) if r else xp,
                      "gp": ops.pl_split(torch.randn(*yp.shape, device=dev)),
                      "dxo": ops.Planes.empty(x, *x.shape), "res": None, "add": None}
                if args.step_like:
                    st["res"] = ops.pl_split(torch.randn(*yp.shape, device=dev))
                    st["add"] = ops.pl_split(torch.randn(*x.shape, device=dev))
                    # the step hands the ReLU mask of the block input over as BITS (written by the producing epilogue)
                    xs = ops.pl_join(st["xp"]).reshape(-1, Cin // 8, 8)
                    st["xp"].mask = ((xs > 0).to(torch.int32) << torch.arange(8, device=dev).to(torch.int32)).sum(-1).to(torch.uint8)
                    del xs
                sets.append(st)
            it = [0]

            def nxt():
                it[0] += 1
                return sets[it[0] % R]

            def f_fwd():
                s_ = nxt()
                ops.pl_conv_fwd_raw(s_["xp"], wp, sc, bi, s_["res"], stride, pad, True, want_mask=args.step_like)

            def f_dgrad():
                s_ = nxt()
                ops.pl_conv_dgrad_raw(s_["gp"], wt, x.shape, k, stride, pad, add=s_["add"], out=s_["dxo"], mask_y=s_["xp"],
                                      mask_scale=None)
            t_f = timeit(f_fwd, iters=12)
            t_d = timeit(f_dgrad, iters=12)
            t_w = float("nan")
            if Cin % 128 == 0 and Cout % 128 == 0:
                dw = torch.zeros_like(w)

                def wgp():
                    s_ = nxt()
                    ops.pl_conv_wgrad_raw(s_["gp"], s_["xp"], w.shape, stride, pad, out=dw)   # (workspace form when it fits)
                t_w = timeit(wgp, iters=12)
            gp = dxo = None
            del sets
            for key, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
                if t == t:
                    tot[key][0] += flop
                    tot[key][1] += t
            print(f"{name:26s} M={n*H*W:7d} N={Cout:4d} K={k*k*Cin:5d}  fwd {flop/t_f/1e9:7.1f}  dgrad {flop/t_d/1e9:7.1f}  "
                  f"wgrad {flop/t_w/1e9:7.1f} TF   ({t_f:.3f} / {t_d:.3f} / {t_w:.3f} ms)")
            del x, w, xp, yp, gp, dxo
            continue
        x = torch.randn(n, H, W, Cin, device=dev)
        w = torch.randn(Cout, k, k, Cin, device=dev) * (Cin * k * k) ** -0.5
        sc, bi = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
        y = ops.conv_fwd_raw(x, w, sc, bi, None, stride, pad, True)
        g = torch.randn_like(y)
        flop = 2.0 * y.numel() * k * k * Cin
        t_f = timeit(lambda: ops.conv_fwd_raw(x, w, sc, bi, None, stride, pad, True))
        t_d = timeit(lambda: ops.conv_dgrad_raw(g, w, x.shape, stride, pad))
        dw = torch.zeros_like(w)

        def wg():
            L.call("stcat_conv_wgrad", g.data_ptr(), x.data_ptr(), dw.data_ptr(), n, H, W, Cin, Cout, k, k, stride,
                   pad, L.stream_of(g))
        t_w = timeit(wg)
        for key, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            tot[key][0] += flop
            tot[key][1] += t
        print(f"{name:26s} M={n*H*W:7d} N={Cout:4d} K={k*k*Cin:5d}  fwd {flop/t_f/1e9:7.1f}  dgrad {flop/t_d/1e9:7.1f}  "
              f"wgrad {flop/t_w/1e9:7.1f} TF   ({t_f:.3f} / {t_d:.3f} / {t_w:.3f} ms)")
        del x, w, y, g, dw
    print("TOTAL " + "  ".join(f"{k} {v[0]/v[1]/1e9:7.1f} TF" for k, v in tot.items() if v[1] > 0))


if __name__ == "__main__":
    main()
