#!/usr/bin/env python3
"""Host vs device time per PHASE of the timed step (harness.TrainStep: the bench's step, headline mode, launch plans ON
by default) — is the GPU starved while the host enqueues the small-kernel phases?

The phases are cut at the composite nodes' boundaries (plans.PlannedFn forward / backward, the same marks as
tools/node_times.py): backbone forward | grounding forward + loss (encoder, decoders, heads, criterion) | grounding
backward | backbone backward + the rest of the step.  `host` = host time between the marks (enqueue), `device` = HIP-event
time between the same marks on the main stream; device >= host means the GPU was the limiter there.

    python tools/phase_times.py [--mma bf16x6p] [--config C3] [--steps 6] [--eager]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib, plans  # noqa: E402
from stcat_amd.harness import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mma", default="bf16x6p")
ap.add_argument("--config", default="C3")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--eager", action="store_true", help="launch plans off")
args = ap.parse_args()

dev = torch.device("cuda:0")
_lib.load()
_lib.set_mma_mode(args.mma)
plans.enable(not args.eager)
ts = TrainStep(dev, args.config)
marks = []
ON = [False]


def mark(name):
    if ON[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, time.perf_counter(), e))


_f, _b = plans.PlannedFn.forward, plans.PlannedFn.backward


def fwd(ctx, node, *a):
    r = _f(ctx, node, *a)
    if "Backbone" in node.__name__:
        mark("backbone forward")
    return r


def bwd(ctx, *g):
    if not any(m[0] == "grounding forward + loss" for m in marks):
        mark("grounding forward + loss")
    if "Backbone" in ctx.node.__name__:
        mark("grounding backward")
    return _b(ctx, *g)


plans.PlannedFn.forward = staticmethod(fwd)
plans.PlannedFn.backward = staticmethod(bwd)

for _ in range(4):
    ts.step()
torch.cuda.synchronize()
ON[0] = True
acc = {}
N = args.steps
for _ in range(N):
    torch.cuda.synchronize()
    marks.clear()
    mark("start")
    ts.step()
    mark("backbone backward + step end")
    torch.cuda.synchronize()
    for (n0, h0, e0), (n1, h1, e1) in zip(marks[:-1], marks[1:]):
        a = acc.setdefault(n1, [0.0, 0.0])
        a[0] += (h1 - h0) * 1e3
        a[1] += e0.elapsed_time(e1)
print(f"# {args.config} {args.mma} plans={'off' if args.eager else 'on'} (plan stats {plans.STATS})")
print("phase: host enqueue ms | device ms between the same marks (device >= host means the GPU was the limiter there)")
th = td = 0.0
for k, (h, d) in acc.items():
    print(f"  {k:32s} host {h/N:6.2f}  device {d/N:6.2f}")
    th += h / N
    td += d / N
print(f"  {'sum':32s} host {th:6.2f}  device {td:6.2f}")
