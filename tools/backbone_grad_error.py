#!/usr/bin/env python3
"""Backbone-only forward + backward with a fixed upstream gradient: weight-gradient error per mode vs the exact-fp32 mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib, synth  # noqa: E402
from stcat_amd.backbone import build_vis_encoder  # noqa: E402

T, res = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 224
dev = torch.device("cuda:0")
_lib.load()
enc = build_vis_encoder(None)
sd = {k[len("vis_encoder."):]: v for k, v in synth.synth_state_dict().items() if k.startswith("vis_encoder.")}
enc.load_state_dict(sd, strict=False)
enc.to(dev).eval()
frames = synth.synth_frames(T, res).to(dev)
res_ = {}
gy = None
for mode in ("f32", "bf16x3", "bf16x3p"):
    _lib.set_mma_mode(mode)
    for p in enc.parameters():
        p.grad = None
    f = enc[0].features_nhwc(frames)
    if gy is None:
        torch.manual_seed(0)
        gy = torch.randn_like(f)
    f.backward(gy)
    res_[mode] = ({n: p.grad.double().clone() for n, p in enc.named_parameters() if p.grad is not None}, f.detach().double())
ref = res_["f32"][0]
for mode in ("bf16x3", "bf16x3p"):
    errs = {n: ((g - ref[n]).norm() / ref[n].norm()).item() for n, g in res_[mode][0].items()}
    by = {}
    for n, e in errs.items():
        fam = [k for k in ("layer2", "layer3", "layer4") if k in n][0]
        by.setdefault(fam, []).append(e)
    print(mode, {k: f"max {max(v):.2e} mean {sum(v)/len(v):.2e}" for k, v in by.items()},
          "feat rel", ((res_[mode][1] - res_["f32"][1]).norm() / res_["f32"][1].norm()).item())
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("   worst:", [(n[-28:], f"{e:.2e}") for n, e in worst])
