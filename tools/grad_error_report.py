#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP path against the fp64 CPU oracle, per contraction mode.

    python tools/grad_error_report.py --config C1 --modes f32,bf16x3,bf16x6 --out gpurun_out/grad_err_C1.json

For every trainable parameter: rel-L2 and max-abs error (relative to the tensor's max) of the HIP gradient
and of the fp32 CPU oracle, both measured against the fp64 oracle run ("exact arithmetic").  This is the
measurement behind the gradient bounds of tests/test_model_parity.py: it says where the
error of a mode is born (family = backbone stage / encoder / decoder) and how it compares with the fp32
reference's own conditioning.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from stcat_amd import synth  # noqa: E402
from tests import test_model_parity as P  # noqa: E402
from tests.backends import use_hip  # noqa: E402


def family(name: str) -> str:
    for k in ("layer2", "layer3", "layer4"):
        if k in name:
            return "backbone." + k
    for k in ("input_proj", "ground_encoder", "ground_decoder.temp_decoder", "ground_decoder.decoder",
              "ground_decoder.template_generator", "temp_embed", "action_embed", "bbox_embed"):
        if name.startswith(k):
            return k
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C1")
    ap.add_argument("--modes", default="f32,bf16x3,bf16x6")
    ap.add_argument("--out", default="")
    ap.add_argument("--no-f64", action="store_true", help="use the fp32 oracle as the yardstick (big configs on small hosts)")
    args = ap.parse_args()
    dev = use_hip()
    T, res, L = synth.CONFIGS[args.config]
    t0 = time.time()
    ref32 = P._run_oracle(T, res, L)
    t32 = time.time() - t0
    g64 = None
    if not args.no_f64:
        t0 = time.time()
        g64 = P._run_oracle(T, res, L, dtype=torch.float64)[4]
        print(f"# oracle fp32 {t32:.1f} s, fp64 {time.time() - t0:.1f} s", flush=True)
    rgrads = ref32[4]
    report = {"config": args.config, "modes": {}}
    for mode in args.modes.split(","):
        keep, losses, grads = P._run_hip(dev, T, res, L, mma=mode)
        fam = {}
        rows = []
        for name, g in rgrads.items():
            hip_name = "ground_decoder.decoder." + name if name.startswith("bbox_embed.") else name
            if hip_name not in grads or name.startswith("ground_decoder.decoder.bbox_embed."):
                continue
            exact = (g64[name] if g64 is not None else g).double()
            a, b = grads[hip_name].double(), g.double()
            # same absolute floors as tests/test_model_parity.py::_compare (exactly-zero gradients: key-side biases)
            nrm = exact.norm().item() + P.GRAD_ABS_FLOOR * exact.numel() ** 0.5 / P.GRAD_TOL
            mx = exact.abs().max().item() + P.GRAD_ABS_FLOOR / P.GRAD_TOL
            e_hip, e_ref = (a - exact).norm().item() / nrm, (b - exact).norm().item() / nrm
            m_hip = (a - exact).abs().max().item() / mx
            rows.append((name, e_hip, e_ref, m_hip))
            f = fam.setdefault(family(name), {"n": 0, "hip_max": 0.0, "ref_max": 0.0, "hip_sum": 0.0, "ref_sum": 0.0,
                                              "gross_max": 0.0})
            f["n"] += 1
            f["hip_max"] = max(f["hip_max"], e_hip)
            f["ref_max"] = max(f["ref_max"], e_ref)
            f["gross_max"] = max(f["gross_max"], m_hip)
            f["hip_sum"] += e_hip
            f["ref_sum"] += e_ref
        out_err = {k: (keep[k].double() - ref32[0][k].double()).abs().max().item()
                   for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights")}
        rows.sort(key=lambda r: -r[1])
        report["modes"][mode] = {
            "families": {k: {"n": v["n"], "hip_relL2_max": v["hip_max"], "hip_relL2_mean": v["hip_sum"] / v["n"],
                             "ref32_relL2_max": v["ref_max"], "ref32_relL2_mean": v["ref_sum"] / v["n"],
                             "hip_maxabs_rel": v["gross_max"]} for k, v in sorted(fam.items())},
            "worst": [{"name": n, "hip": h, "ref32": r, "maxabs": m} for n, h, r, m in rows[:12]],
            "output_max_abs_err_vs_fp32_oracle": out_err,
            "loss_total": losses["total"], "loss_total_oracle": ref32[3]["total"],
            "span": keep["post_sted"], "span_oracle": [ref32[2]],
        }
        print(f"== {mode}: outputs {out_err}")
        for k, v in report["modes"][mode]["families"].items():
            print(f"  {k:36s} n={v['n']:3d} hip relL2 max {v['hip_relL2_max']:.2e} mean {v['hip_relL2_mean']:.2e} | "
                  f"fp32-oracle max {v['ref32_relL2_max']:.2e} mean {v['ref32_relL2_mean']:.2e} | max-abs {v['hip_maxabs_rel']:.2e}",
                  flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
