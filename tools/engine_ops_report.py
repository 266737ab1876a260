#!/usr/bin/env python3
"""GPU box: which aten ops run inside the autograd engine thread during one C3 training step (torch.profiler sees every
thread), grouped by the autograd node that issued them — AccumulateGrad clones, materialised zero gradients, sums."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib, ops, synth  # noqa: E402
from stcat_amd.misc import BoxList, NestedTensor  # noqa: E402
from stcat_amd.pipeline import SyntheticText, build_model  # noqa: E402

dev = torch.device("cuda:0")
_lib.load()
_lib.set_mma_mode("bf16x3p")
T, res, L = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C1"]
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
    criterion.weighted_total(wd).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
WATCH = ("aten::clone", "aten::copy_", "aten::zeros", "aten::zero_", "aten::zeros_like", "aten::fill_", "aten::add", "aten::add_",
         "aten::sum", "aten::cat", "aten::contiguous", "aten::_to_copy", "aten::mul", "aten::empty_like", "aten::ones_like")
agg = collections.Counter()
for e in prof.events():
    if e.name not in WATCH:
        continue
    par = e.cpu_parent
    chain = []
    nested = False
    while par is not None:
        if par.name in WATCH:
            nested = True
            break
        chain.append(par.name)
        par = par.cpu_parent
    if nested:
        continue
    top = next((c for c in chain if "autograd::engine" in c or "Backward" in c or "Fn" in c), chain[0] if chain else "(top level)")
    agg[(top[:90], e.name)] += 1
kern = collections.Counter()
for e in prof.events():
    if e.device_type is not None and "cuda" in str(e.device_type).lower():
        kern[e.name.split("(")[0][:60]] += 1
print("aten ops by issuing node:")
for (top, name), c in sorted(agg.items(), key=lambda kv: -kv[1])[:40]:
    print(f"  {c:4d}  {name:18s} {top}")
print("device activities that are not ours:")
for k, c in sorted(kern.items(), key=lambda kv: -kv[1]):
    if not any(s in k for s in ("igemm", "ew_kernel", "ew2d", "layernorm", "dropout", "mha_", "attn_", "act_bwd", "planes", "stg_loss",
                                "weight_", "sine", "colsum", "small_linear", "maxpool", "grad_", "optim", "map2d", "rowscale",
                                "temporal", "pos_sine")):
        print(f"  {c:4d}  {k}")
