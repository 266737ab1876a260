#!/usr/bin/env python3
"""Relative error of the backbone's layer4 features (forward) per contraction mode, against the exact-fp32 mode."""
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
feats = {}
with torch.no_grad():
    for mode in ("f32", "bf16x6", "bf16x3", "bf16x3p"):
        _lib.set_mma_mode(mode)
        feats[mode] = enc[0].features_nhwc(frames).double()
ref = feats["f32"]
for mode, f in feats.items():
    print(f"{mode:8s} rel-L2 {((f - ref).norm() / ref.norm()).item():.3e}  max-abs {(f - ref).abs().max().item():.3e}  (|ref|max {ref.abs().max().item():.3g})")
