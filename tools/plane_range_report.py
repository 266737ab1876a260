#!/usr/bin/env python3
"""Magnitudes of every plane-format tensor of one C3 step (activations, weights, gradients): would fp16 hi + lo planes
 need a scale per TENSOR, or do three constants (activations, weights, one loss scale) do?

fp16 keeps 22 bits in two planes only while the value's lower plane stays normal: |x| >= 2^-3 or so (the lower plane is
2^-11 |x| and fp16's normal range ends at 2^-14), and overflows above 65504.  Prints, per tensor class, the spread of
log2(rms) and log2(amax) over all launches of a step, and the share of each tensor's energy (sum x^2) carried by elements
below a threshold.  Usage (GPU box): python tools/plane_range_report.py [--config C3]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib as L, ops  # noqa: E402
from stcat_amd.harness import TrainStep  # noqa: E402


def stats(p):
    x = ops.pl_join(p) if isinstance(p, ops.Planes) else p.float()
    x = x.reshape(-1)
    if x.numel() > (1 << 24):
        x = x[:: x.numel() // (1 << 24)]
    a = x.abs()
    rms = a.square().mean().sqrt().item()
    return rms, a.max().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    args = ap.parse_args()
    L.load()
    L.set_mma_mode("bf16x6p")
    dev = torch.device("cuda:0")
    ts = TrainStep(dev, args.config, train=True)
    ts.step()
    rec = {"activation (conv input)": [], "weight": [], "gradient dY (dgrad / wgrad input)": [], "conv output (planes)": [],
           "data gradient dX (planes out)": []}
    f0, d0, w0 = ops.pl_conv_fwd_raw, ops.pl_conv_dgrad_raw, ops.pl_conv_wgrad_raw

    def fwd(x, w, *a, **k):
        rec["activation (conv input)"].append(stats(x))
        rec["weight"].append(stats(w))
        r = f0(x, w, *a, **k)
        if r[0] is not None:
            rec["conv output (planes)"].append(stats(r[0]))
        return r

    def dgrad(g, wt, *a, **k):
        rec["gradient dY (dgrad / wgrad input)"].append(stats(g))
        r = d0(g, wt, *a, **k)
        rec["data gradient dX (planes out)"].append(stats(r[0] if isinstance(r, tuple) else r))
        return r

    ops.pl_conv_fwd_raw, ops.pl_conv_dgrad_raw = fwd, dgrad
    import stcat_amd.backbone as bb
    try:
        with ops.single_stream():
            ts.step()
        torch.cuda.synchronize()
    finally:
        ops.pl_conv_fwd_raw, ops.pl_conv_dgrad_raw, ops.pl_conv_wgrad_raw = f0, d0, w0
    print(f"# {args.config}, one train-mode step, mode bf16x6p; per class: launches, log2(rms) min / median / max, log2(amax) min / max")
    for k, v in rec.items():
        if not v:
            continue
        lr = sorted(math.log2(max(r, 1e-45)) for r, _ in v)
        la = sorted(math.log2(max(a, 1e-45)) for _, a in v)
        print(f"{k:38s} n={len(v):4d}  log2 rms {lr[0]:7.1f} / {lr[len(lr) // 2]:7.1f} / {lr[-1]:7.1f}   log2 amax {la[0]:7.1f} / {la[-1]:7.1f}")
    print("# fp16 hi + lo needs log2 rms >= about -3 for 22 bits and log2 amax < 16; a class whose spread (max - min of log2 rms)\n"
          "# is under ~12 can take ONE power-of-two scale; a wider class needs a scale per tensor.")


if __name__ == "__main__":
    main()
