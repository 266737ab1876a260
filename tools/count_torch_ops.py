#!/usr/bin/env python3
"""Which PyTorch (non-stcat) ops does one training step still issue, and from where?  (CPU-side op profile with
Python stacks; forward call sites are attributed directly, backward ones show up as autograd node names.)"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib, ops, synth
from stcat_amd.misc import BoxList, NestedTensor
from stcat_amd.pipeline import SyntheticText, build_model
_lib.load(); _lib.set_mma_mode("bf16x3")
dev = torch.device("cuda:0")
T, res, L = 8, 224, 10
model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
model.train(); synth.fill_module_(model); model.to(dev)
arena = ops.enable_zero_arena(dev, 120_000_000)
frames = synth.synth_frames(T, res).to(dev); mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
act, tb = synth.synth_targets(T)
targets = [{"actioness": act.to(dev), "boxs": BoxList(tb).to(dev)}]
plan = criterion.plan(targets, [T], dev)
def step():
    for p in model.parameters(): p.grad = None
    arena.reset()
    out = model(NestedTensor(frames, mask, [T]), ["synthetic"])
    criterion(out, targets, [T], plan=plan)
    criterion.weighted_total(wd).backward()
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=cfg) as prof:
    step()
torch.cuda.synchronize()
names = ("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::cat", "aten::zero_", "aten::clone", "aten::mul", "aten::sum", "aten::select_backward", "aten::slice_backward", "aten::index", "aten::stack")
cnt = collections.Counter(); where = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name in names:
        cnt[ev.name] += 1
        st = [f for f in (ev.stack or []) if "stcat_amd" in f or "tools/" in f]
        where[ev.name][(st[0].split("/")[-1][:60] if st else "(autograd / no python frame)")] += 1
for k, v in cnt.most_common():
    print(k, v, where[k].most_common(8))
