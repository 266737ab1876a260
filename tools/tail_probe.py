#!/usr/bin/env python3
"""How long does the main stream wait for the weight-gradient stream at every join of a step (eager launches, C3,
bf16x6p)?  The last join — the end of the backbone's backward — is the step's tail: a long wait there means the
weight-gradient stream, not the data-gradient chain, ends the step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib, ops, plans  # noqa: E402
from stcat_amd.harness import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
_lib.load()
_lib.set_mma_mode("bf16x6p")
plans.enable(False)
ts = TrainStep(dev, "C3", pipeline_prefix=True)
ON = [False]
marks = []
_join = ops.WgradStream.join


def join(self, *outputs):
    if ON[0] and self.active:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(self.main)
        _join(self, *outputs)
        e1.record(self.main)
        marks.append((e0, e1))
    else:
        _join(self, *outputs)


ops.WgradStream.join = join
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
ON[0] = True
acc = None
for _ in range(5):
    marks.clear()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    ts.step()
    s1.record()
    torch.cuda.synchronize()
    row = [s0.elapsed_time(s1)] + [a.elapsed_time(b) for a, b in marks] + [s0.elapsed_time(marks[-1][0])]
    acc = row if acc is None else [x + y for x, y in zip(acc, row)]
acc = [x / 5 for x in acc]
print(f"step {acc[0]:.2f} ms (eager launches); {len(acc) - 2} joins; wait of the main stream at each join (ms): "
      + " ".join(f"{x:.2f}" for x in acc[1:-1]) + f"; the last join begins {acc[-1]:.2f} ms into the step")
