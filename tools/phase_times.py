#!/usr/bin/env python3
"""Host vs device time per phase of the step (is the GPU starved while the host enqueues the small-kernel phases?)."""
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
_lib.set_mma_mode(sys.argv[1] if len(sys.argv) > 1 else "bf16x6p")
T, res, L = synth.CONFIGS["C3"]
model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
model.train()
synth.fill_module_(model)
model.to(dev)
arena = ops.enable_zero_arena(dev, 120_000_000)
frames = synth.synth_frames(T, res).to(dev)
mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
act, tb = synth.synth_targets(T)
targets = [{"actioness": act.to(dev), "boxs": BoxList(tb).to(dev)}]
plan = criterion.plan(targets, [T], dev)
plan.num_boxes(dev)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, time.perf_counter(), e))


orig = model.vis_encoder.forward_tokens


def ft(fr, m):
    out = orig(fr, m)
    mark("backbone fwd enqueued")
    feat = out[0]
    if feat.requires_grad:
        feat.register_hook(lambda g: (mark("non-backbone bwd enqueued"), g)[1])
    return out


model.vis_encoder.forward_tokens = ft


def step():
    marks.clear()
    for p in model.parameters():
        p.grad = None
    arena.reset()
    mark("start")
    out = model(NestedTensor(frames, mask, [T]), ["synthetic"])
    criterion(out, targets, [T], plan=plan)
    total = criterion.weighted_total(wd)
    mark("fwd + loss enqueued")
    total.backward()
    mark("bwd enqueued")


for _ in range(4):
    step()
torch.cuda.synchronize()
acc = {}
N = 6
for _ in range(N):
    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    for (n0, h0, e0), (n1, h1, e1) in zip(marks[:-1], marks[1:]):
        a = acc.setdefault(n1, [0.0, 0.0])
        a[0] += (h1 - h0) * 1e3
        a[1] += e0.elapsed_time(e1)
print("phase: host enqueue ms | device ms between the same marks (device >= host means the GPU was the limiter there)")
for k, (h, d) in acc.items():
    print(f"  {k:28s} host {h/N:6.2f}  device {d/N:6.2f}")
