"""The synthetic training step — counterpart of the reference's inner loop (scripts/train_net.py:97-143, without the
optimizer) on one synthetic video per rank: zero_grad, forward, VideoSTGLoss, weighted sum, backward, gradient exchange.

`bench.py` times exactly `TrainStep.step()`, and `tests/test_model_parity.py` compares exactly that — the launch plans
replayed, the zero arena, the bucketed reducer, the loss plan — with the reference's fixtures ("the
path the bench times is not the path the oracle tests check").
"""
from __future__ import annotations

import torch

from . import ops, plans, synth
from .dist import GradBucketReducer
from .misc import BoxList, NestedTensor
from .pipeline import SyntheticText, build_model


class TrainStep:
    """model + criterion + reducer + the per-step host sequence.

    frames / mask / targets default to the benchmark's synthetic clip of `config` (seeded per rank); a test passes the
    clip of a fixture instead.  `loss_plan_inside=True` (default): the loss's target-derived index tensors and its
    1-element box-count all-reduce (criterion.py:160-192, 175-178) are rebuilt inside EVERY step, as the reference does;
    the targets stay on the host side of the input pipeline, so building them costs no device -> host sync."""

    def __init__(self, dev, config="C3", rank: int = 0, train: bool = True, roberta_dummy: bool = False,
                 force_comm: bool = False, clip=None, targets=None, loss_plan_inside: bool = True, seed: int = 20260929,
                 arena_elems: int = 120_000_000, clips=None, pipeline_prefix: bool = False):
        self.dev = dev
        T, res, L = synth.CONFIGS[config] if isinstance(config, str) else config
        self.T, self.res, self.L = T, res, L
        self.model, self.criterion, self.wd = build_model(None, SyntheticText(synth.synth_text(L)))
        if train:
            self.model.train()       # the measured workload: dropout active (FrozenBN has no train-mode state)
            ops.manual_seed(seed, rank)
        else:
            self.model.eval()        # dropout off (the parity configuration); gradients flow
        synth.fill_module_(self.model)
        self.model.to(dev)
        self.reducer = GradBucketReducer(self.model, extra_numel=124_645_632 if roberta_dummy else 0,
                                         force_comm=force_comm)
        self.arena = ops.enable_zero_arena(dev, arena_elems)   # weight-gradient accumulators etc.: one memset per step
        if clip is None:
            frames = synth.synth_frames(T, res, seed=1000 * 3 + rank)
            mask = torch.zeros(T, res, res, dtype=torch.bool)
        else:
            frames, mask = clip
        # `clips`: several (frames, mask) of one geometry, visited round-robin (step k runs clips[k % len]); with
        # `pipeline_prefix` each step declares the NEXT step's frames to the backbone (Backbone.stage_next), which computes
        # their frozen prefix under this step's grounding section — what a training loop with a look-ahead loader does
        if clips is None:
            clips = [(frames, mask)]
        self.clips = [NestedTensor(f.to(dev), m.to(dev), [T]) for f, m in clips]
        self.clip_index = 0
        self.pipeline_prefix = pipeline_prefix
        self.videos = self.clips[0]
        if targets is None:
            act, tb = synth.synth_targets(T, seed=rank)
            targets = [{"actioness": act, "boxs": BoxList(tb)}]
        self.targets_host = targets                      # annotations as the loader holds them (host memory)
        self.targets = [{"actioness": t["actioness"].to(dev), "boxs": t["boxs"].to(dev)} for t in targets]
        self.loss_plan_inside = loss_plan_inside
        self._plan = None
        self.uniform_w = len({self.wd[k] for k in self.wd if k.startswith("loss_bbox")}) == 1
        self.last_out = None
        self.keep_outputs = False

    # ---- the loss's target-only tensors ---------------------------------------------------------------------------
    def loss_plan(self):
        if self.loss_plan_inside or self._plan is None:
            self._plan = self.criterion.plan(self.targets_host, [self.T], self.dev)
            self._plan.num_boxes(self.dev)               # criterion.py:175-178: the 1-element all-reduce (no host sync)
        return self._plan

    # ---- one video: forward + loss + backward; no host syncs ------------------------------------------------------
    def compute(self):
        dev = self.dev
        ops.dropout_begin_step(dev)
        self.arena.reset()
        # training updates the fp32 weights between steps, so the per-step split of all conv weights into bf16 planes
        # (+ the transposed, FrozenBN-scaled copies for the data gradients) is part of every step: no optimizer runs
        # inside the timed region, hence the epoch bump that makes the refresh launch run as it does in training
        ops.WEIGHT_EPOCH += 1
        plan = self.loss_plan()
        self.videos = self.clips[self.clip_index % len(self.clips)]
        self.clip_index += 1
        if self.pipeline_prefix:
            nxt = self.clips[self.clip_index % len(self.clips)]
            self.model.vis_encoder[0].stage_next(nxt.tensors)
        out = self.model(self.videos, ["synthetic"])
        if self.keep_outputs:        # (tests: the criterion overwrites pred_boxes with the GT-span rows, criterion.py:168-171)
            keys = ("pred_boxes", "pred_sted", "pred_actioness", "weights")
            self.last_out = {k: out[k].detach().clone() for k in keys}
            self.last_out["aux"] = [{k: a[k].detach().clone() for k in keys} for a in out["aux_outputs"]]
        losses = self.criterion(out, self.targets, [self.T], plan=plan)
        self.last_losses = losses
        total = self.criterion.weighted_total(self.wd) if self.uniform_w else sum(losses[k] * self.wd[k] for k in losses)
        total.backward()
        return total

    def step(self):
        self.reducer.zero_grad()
        total = self.compute()
        self.reducer.finish()
        return total

    def gradients(self):
        return {n: p.grad.detach() for n, p in self.model.named_parameters() if p.grad is not None}

    def close(self):
        """undo the process-wide state a step object installs (tests build several in one process)"""
        self.reducer.close()
        ops.disable_zero_arena()
        plans.clear()
