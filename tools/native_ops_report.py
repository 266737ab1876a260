#!/usr/bin/env python3
"""Which torch (aten) ops does one training step still run, and from where?  Runs the tiny clip through the host
emulator build under torch.profiler (CPU, with stacks) and lists the aten ops that would be device launches on the GPU
box, by the innermost stcat_amd frame — the work-list for 'zero at::native kernels in the step'."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from backends import use_emu, use_hip  # noqa: E402
from stcat_amd import ops, synth  # noqa: E402
from stcat_amd.misc import BoxList, NestedTensor  # noqa: E402
from stcat_amd.pipeline import SyntheticText, build_model  # noqa: E402

HIP = "--hip" in sys.argv          # on the GPU box: the real library, config C1, zero arena on (as bench.py runs)
dev = use_hip() if HIP else use_emu()
T, res, L = (8, 224, 10) if HIP else (2, 64, 3)
model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
synth.fill_module_(model)
model.to(dev).train()
frames = synth.synth_frames(T, res).to(dev)
videos = NestedTensor(frames, torch.zeros(T, res, res, dtype=torch.bool, device=dev), [T])
act, tb = synth.synth_targets(T)
targets = [{"actioness": act.to(dev), "boxs": BoxList(tb, (res, res)).to(dev)}]
plan = criterion.plan(targets, [T], dev)
plan.num_boxes(dev)
arena = ops.enable_zero_arena(dev, 120_000_000) if HIP else None


def step():
    for p in model.parameters():
        p.grad = None
    if arena is not None:
        arena.reset()
    out = model(videos, ["synthetic"])
    criterion(out, targets, [T], plan=plan)
    total = criterion.weighted_total(wd)
    total.backward()


step()
import traceback  # noqa: E402

from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEW = ("view", "reshape", "expand", "permute", "transpose", "t.default", "slice", "select", "unsqueeze", "squeeze", "detach",
        "alias", "as_strided", "empty", "unbind", "split", "narrow", "_unsafe_view", "unflatten", "flatten", "sym_", "is_",
        "_local_scalar_dense", "lift_fresh", "stride", "size", "numel", "dim", "_reshape_alias", "chunk", "contiguous")
agg = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        short = name.replace("aten.", "")
        if not any(short.startswith(v) for v in VIEW) and (HIP or not (short.startswith("zero_") or short.startswith("zeros"))):
            where = "autograd engine (no stcat_amd frame)"
            for fr in reversed(traceback.extract_stack()):
                if "stcat_amd/" in fr.filename:
                    where = f"{fr.filename.split('stcat_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            agg[(where, short)] += 1
        return func(*args, **(kwargs or {}))


with Rec():
    step()
tot = sum(agg.values())
print(f"{tot} launching aten ops in one train step (tiny clip; zero fills excluded: the zero arena serves them); by call site:")
for (where, name), c in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"  {c:4d}  {name:34s} {where}")
