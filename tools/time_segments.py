#!/usr/bin/env python3
"""Wall/GPU time of the step's segments at C3 (events on the compute stream)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib, ops, synth
from stcat_amd.misc import BoxList, NestedTensor
from stcat_amd.pipeline import SyntheticText, build_model

_lib.load(); _lib.set_mma_mode(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
dev = torch.device("cuda:0")
T, res, L = synth.CONFIGS["C3"]
model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
model.eval(); synth.fill_module_(model); model.to(dev)
arena = ops.enable_zero_arena(dev, 120_000_000)
frames = synth.synth_frames(T, res).to(dev)
mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
act, tb = synth.synth_targets(T)
targets = [{"actioness": act.to(dev), "boxs": BoxList(tb).to(dev)}]

plan = criterion.plan(targets, [T], dev)

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

def step(log=False):
    for p in model.parameters(): p.grad = None
    arena.reset()
    marks = [("start", ev(), time.perf_counter())]
    feat, m, vis_pos = model.vis_encoder.forward_tokens(frames, mask)
    marks.append(("backbone fwd", ev(), time.perf_counter()))
    n, h, w, c = feat.shape
    vis = ops.linear(feat.view(n * h * w, c), model.input_proj.weight.view(-1, c), model.input_proj.bias)
    (tm, tmem, _), tcls = model.text_encoder(None, dev)
    memory, mem_mask, fcls, vcls, mem_pos = model.ground_encoder.run(vis.view(n, h * w, -1), m.flatten(1), vis_pos, tm, tmem)
    marks.append(("input_proj + encoder fwd", ev(), time.perf_counter()))
    hs, ref, time_hs, weights, _ = model.ground_decoder.run(memory.contiguous(), mem_mask, mem_pos, fcls, vcls)
    marks.append(("decoders fwd", ev(), time.perf_counter()))
    out = {"weights": weights[-1]}
    coord = ops.sigmoid(ops.add(model.bbox_embed(hs), ops.inverse_sigmoid(ref)))
    out["pred_boxes"] = coord[-1]
    sted = model.temp_embed(time_hs)[:, None]; out["pred_sted"] = sted[-1]
    actn = model.action_embed(time_hs)[:, None]; out["pred_actioness"] = actn[-1]
    out["aux_outputs"] = [{"pred_sted": sted[i], "pred_boxes": coord[i], "weights": weights[i], "pred_actioness": actn[i]} for i in range(5)]
    losses = criterion(out, targets, [T], plan=plan)
    total = criterion.weighted_total(wd)
    marks.append(("heads + loss", ev(), time.perf_counter()))
    total.backward()
    marks.append(("backward (all)", ev(), time.perf_counter()))
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    if log:
        for (n0, e0, h0), (n1, e1, h1) in zip(marks, marks[1:]):
            print(f"{n1:28s} gpu {e0.elapsed_time(e1):8.2f} ms   host-enqueue {1e3*(h1-h0):8.2f} ms")
        print(f"{'total':28s} gpu {marks[0][1].elapsed_time(marks[-1][1]):8.2f} ms   host {1e3*(t_end-marks[0][2]):8.2f} ms")

for _ in range(3): step()
step(log=True)
# backward split: hooks are awkward; instead time backward of the grounding part alone by detaching the backbone
