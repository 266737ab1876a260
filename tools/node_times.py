#!/usr/bin/env python3
"""Untraced node-level timeline of the timed step (harness.TrainStep, headline mode, launch plans on).

Every composite node's forward / backward (plans.PlannedFn) is bracketed by an event on the stream it runs on and a host
timestamp: per node, when the HOST started / finished enqueuing it and when the DEVICE started / finished executing it,
in ms from the step's first event.  `host lead` = device start - host enqueue end of the same node: negative means the
GPU waited for the host there (launch-bound), positive means the launches were queued ahead.

    python tools/node_times.py [--mma bf16x6p] [--config C3] [--steps 6]
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
ap.add_argument("--no-prefix-pipeline", action="store_true", help="every step computes its clip's frozen prefix itself")
ap.add_argument("--force-comm", action="store_true", help="the complete RCCL path with ONE rank (bench.py's STCAT_FORCE_COMM=1)")
args = ap.parse_args()

dev = torch.device("cuda:0")
_lib.load()
_lib.set_mma_mode(args.mma)
plans.enable(not args.eager)
if args.force_comm:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    from stcat_amd.dist import init_rccl_process_group
    init_rccl_process_group(dev)
ts = TrainStep(dev, args.config, pipeline_prefix=not args.no_prefix_pipeline, force_comm=args.force_comm)
marks = []
ON = [False]


def mark(name):
    if ON[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, time.perf_counter(), e))


_f, _b = plans.PlannedFn.forward, plans.PlannedFn.backward


def fwd(ctx, node, *a):
    mark(("fwd", node.__name__, 0))
    r = _f(ctx, node, *a)
    mark(("fwd", node.__name__, 1))
    return r


def bwd(ctx, *g):
    n = ctx.node.__name__
    mark(("bwd", n, 0))
    r = _b(ctx, *g)
    mark(("bwd", n, 1))
    return r


plans.PlannedFn.forward = staticmethod(fwd)
plans.PlannedFn.backward = staticmethod(bwd)

# the next clip's frozen prefix (Backbone._fill): events on ITS stream around the launches
from stcat_amd import backbone as _bb  # noqa: E402
_pf = _bb._prefix_forward
PREFIX = []


def prefix_forward(frames, body, out):
    if ON[0]:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    r = _pf(frames, body, out)
    if ON[0]:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PREFIX.append((e0, e1))
    return r


_bb._prefix_forward = prefix_forward

for _ in range(4):
    ts.step()
torch.cuda.synchronize()
ON[0] = True
rows = {}
wall = []
pre_rows = [0.0, 0.0, 0]
for _ in range(args.steps):
    torch.cuda.synchronize()
    marks.clear()
    PREFIX.clear()
    t0 = time.perf_counter()
    mark(("step", "start", 0))
    ts.step()
    mark(("step", "end", 1))
    torch.cuda.synchronize()
    wall.append((time.perf_counter() - t0) * 1e3)
    e0, h0 = marks[0][2], marks[0][1]
    for a_, b_ in PREFIX:
        pre_rows[0] += e0.elapsed_time(a_)
        pre_rows[1] += e0.elapsed_time(b_)
        pre_rows[2] += 1
    for name, h, e in marks:
        r = rows.setdefault(name, [0.0, 0.0, 0])
        r[0] += (h - h0) * 1e3
        r[1] += e0.elapsed_time(e)
        r[2] += 1
print(f"# {args.config} {args.mma} plans={'off' if args.eager else 'on'}: wall {sum(wall) / len(wall):.2f} ms/step "
      f"(with the marks), plan stats {plans.STATS}")
print("# node                      host: enq start   enq end | device: start      end   (dur) | host lead at start")
if pre_rows[2]:
    a_, b_ = pre_rows[0] / pre_rows[2], pre_rows[1] / pre_rows[2]
    print(f"  next clip's frozen prefix (side stream)                  | {a_:9.2f} {b_:9.2f} ({b_ - a_:6.2f}) |")
keys = [k for k in rows if k[2] == 0]
for k in keys:
    k1 = (k[0], k[1], 1)
    if k1 not in rows:
        continue
    a, b = rows[k], rows[k1]
    hs, ds = a[0] / a[2], a[1] / a[2]
    he, de = b[0] / b[2], b[1] / b[2]
    print(f"  {k[0]} {k[1]:22s} {hs:9.2f} {he:9.2f} | {ds:9.2f} {de:9.2f} ({de - ds:6.2f}) | {ds - hs:7.2f}")
