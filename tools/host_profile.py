#!/usr/bin/env python3
"""cProfile of the host side of a training step (python bench-like loop) — where the enqueue time goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib, ops, plans, synth  # noqa: E402
from stcat_amd.misc import BoxList, NestedTensor  # noqa: E402
from stcat_amd.pipeline import SyntheticText, build_model  # noqa: E402

dev = torch.device("cuda:0")
_lib.load()
_lib.set_mma_mode(sys.argv[1] if len(sys.argv) > 1 else "bf16x6p")
T, res, L = synth.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C3"]
plans.enable(not (len(sys.argv) > 3 and sys.argv[3] == "noplans"))
model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
model.train()
synth.fill_module_(model)
model.to(dev)
arena = ops.enable_zero_arena(dev, 120_000_000)
frames = synth.synth_frames(T, res).to(dev)
videos = NestedTensor(frames, torch.zeros(T, res, res, dtype=torch.bool, device=dev), [T])
act, tb = synth.synth_targets(T)
targets = [{"actioness": act.to(dev), "boxs": BoxList(tb).to(dev)}]
plan = criterion.plan(targets, [T], dev)
plan.num_boxes(dev)


def step():
    for p in model.parameters():
        p.grad = None
    arena.reset()
    out = model(videos, ["synthetic"])
    criterion(out, targets, [T], plan=plan)
    total = criterion.weighted_total(wd)
    total.backward()


for _ in range(4):
    step()
torch.cuda.synchronize()
plans.STATS["run_s"] = 0.0
t0 = time.perf_counter()
for _ in range(5):
    step()
host = (time.perf_counter() - t0) / 5
torch.cuda.synchronize()
print(f"host enqueue {1e3*host:.1f} ms/step, wall {(time.perf_counter()-t0)/5*1e3:.1f} ms/step, inside Plan.run "
      f"{1e3*plans.STATS['run_s']/5:.2f} ms/step, plans {plans.STATS}")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
