#!/usr/bin/env python3
"""eager vs eager vs plan-replayed steps at C1: per step, the worst per-tensor rel-L2 gradient difference (run-to-run noise
of two eager runs next to eager-vs-replay)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_plans as tp  # noqa: E402
from tests.backends import use_hip  # noqa: E402

dev = use_hip()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mma = sys.argv[2] if len(sys.argv) > 2 else "bf16x6p"
A, _ = tp._run(dev, 8, 224, steps, False, mma=mma)
B, _ = tp._run(dev, 8, 224, steps, False, mma=mma)
C, st = tp._run(dev, 8, 224, steps, True, mma=mma)
print(st)


def worst(x, y):
    out = []
    for n in x[2]:
        a, b = x[2][n].double(), y[2][n].double()
        e = (a - b).norm().item() / (b.norm().item() + 2e-6 * b.numel() ** 0.5 / 3e-3)
        out.append((e, n))
    out.sort(reverse=True)
    return out[:3]


for k in range(steps):
    print(k, "eager/eager", [(f"{e:.1e}", n[-40:]) for e, n in worst(B[k], A[k])])
    print(k, "plan /eager", [(f"{e:.1e}", n[-40:]) for e, n in worst(C[k], A[k])])
