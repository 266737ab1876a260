#!/usr/bin/env python3
"""GPU box: C3 training step fed with resident fp32 NCHW frames vs resident uint8 HWC frames (no H2D), and per-phase
host times of the uint8 step — where the uint8 input path loses time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib, ops, synth  # noqa: E402
from stcat_amd.misc import BoxList, NestedTensor  # noqa: E402
from stcat_amd.pipeline import SyntheticText, build_model  # noqa: E402

dev = torch.device("cuda:0")
_lib.load()
_lib.set_mma_mode("bf16x3p")
T, res, L = synth.CONFIGS["C3"]
model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
model.train()
synth.fill_module_(model)
model.to(dev)
arena = ops.enable_zero_arena(dev, 120_000_000)
f32 = synth.synth_frames(T, res).to(dev)
u8 = torch.randint(0, 256, (T, res, res, 3), dtype=torch.uint8, device=dev)
mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
act, tb = synth.synth_targets(T)
targets = [{"actioness": act.to(dev), "boxs": BoxList(tb).to(dev)}]
plan = criterion.plan(targets, [T], dev)
plan.num_boxes(dev)


def step(frames):
    for p in model.parameters():
        p.grad = None
    ops.dropout_begin_step(dev)
    arena.reset()
    out = model(NestedTensor(frames, mask, [T]), ["synthetic"])
    criterion(out, targets, [T], plan=plan)
    criterion.weighted_total(wd).backward()


for name, fr in (("fp32", f32), ("uint8", u8), ("fp32", f32), ("uint8", u8)):
    for _ in range(2):
        step(fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step(fr)
    h = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms/step (host enqueue {1e3 * h / 5:.2f})")
# with the upload from pinned host memory inside the step (resident device buffer, async copy)
for name, fr in (("fp32+h2d", f32), ("uint8+h2d", u8), ("fp32+h2d", f32), ("uint8+h2d", u8)):
    host = fr.cpu().pin_memory()
    buf = torch.empty_like(fr)
    for _ in range(2):
        step(buf.copy_(host, non_blocking=True))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step(buf.copy_(host, non_blocking=True))
    h = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms/step (host enqueue {1e3 * h / 5:.2f})")
# forward only, synced phases
for name, fr in (("fp32", f32), ("uint8", u8)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        feat = model.vis_encoder[0].features_nhwc(fr)
    torch.cuda.synchronize()
    print(f"{name}: backbone forward alone {1e3 * (time.perf_counter() - t0):.2f} ms")
