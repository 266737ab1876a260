"""Standalone wiring of the hot path (the GPU box has no reference checkout): counterparts of
``STCATNet`` (models/pipeline.py:12-121), ``VideoSTGLoss`` (models/criterion.py:11-208),
``PostProcess`` (models/post_processor.py:13-55) and ``build_model`` (models/__init__.py:5-41), with the
same call signatures and output dictionaries.  In drop-in mode (``stcat_amd.install()``) the reference's own
``pipeline.py`` / ``criterion.py`` / ``post_processor.py`` are used unchanged and only the three factories
are rebound.

The text encoder is outside the hot path: ``text_encoder`` is any callable
``(texts, device) -> ((mask[b,L] bool, memory[L,b,256], tokens), cls[b,256])`` — the reference's Roberta
module satisfies it; the synthetic harness passes pre-computed boundary tensors.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import composite, ops, plans
from .backbone import build_vis_encoder
from .grounding import MLP, build_decoder, build_encoder
from .misc import NestedTensor


class SyntheticText(nn.Module):
    """Returns fixed boundary tensors (stands in for language_model/bert.py:59-74 in the synthetic harness)."""

    def __init__(self, text):
        super().__init__()
        (mask, mem, _), cls = text
        self.register_buffer("mask", mask, persistent=False)
        self.register_buffer("mem", mem, persistent=False)
        self.register_buffer("cls", cls, persistent=False)

    def forward(self, texts, device):
        return (self.mask, self.mem, None), self.cls


class STCATNet(plans.InvalidatesPlans, nn.Module):
    def __init__(self, cfg=None, text_encoder: Optional[nn.Module] = None):
        super().__init__()
        self.use_attn = True if cfg is None else cfg.SOLVER.USE_ATTN
        self.use_aux_loss = True if cfg is None else cfg.SOLVER.USE_AUX_LOSS
        self.use_actioness = True if cfg is None else cfg.MODEL.STCAT.USE_ACTION
        self.query_dim = 4 if cfg is None else cfg.MODEL.STCAT.QUERY_DIM
        if not self.use_attn:
            raise ValueError("SOLVER.USE_ATTN=False is not runnable in the reference either (pipeline.py:99)")
        self.vis_encoder = build_vis_encoder(cfg)
        self.text_encoder = text_encoder
        self.ground_encoder = build_encoder(cfg)
        self.ground_decoder = build_decoder(cfg)
        hidden = 256
        self.input_proj = nn.Conv2d(self.vis_encoder.num_channels, hidden, kernel_size=1)  # parameter container
        self.temp_embed = MLP(hidden, hidden, 2, 2, dropout=0.3)
        self.bbox_embed = MLP(hidden, hidden, 4, 3)
        self.action_embed = MLP(hidden, hidden, 1, 2, dropout=0.3) if self.use_actioness else None
        self.ground_decoder.decoder.bbox_embed = self.bbox_embed                             # pipeline.py:50

    _graphed_decoder = None

    def capture_decoder(self, on: bool = True) -> None:
        """BASELINE.json configs[4] ("hipGraph-captured decoder"): under `torch.no_grad()` the template generator, the six
        box-decoder layers with their anchor refinement and box head, the six time-decoder layers on the forked stream and
        the span / actioness heads (query_decoder.py:13-147, 169-247, 310-438, 587-660; pipeline.py:77-103) — ~700
        launches on [T,256] states, each shorter than its launch latency — are captured ONCE per input shape and replayed
        with one hipGraphLaunch (stcat_amd/graph.py: GraphedDecoder).  Training steps are untouched (launch plans)."""
        from .graph import GraphedDecoder
        self._graphed_decoder = GraphedDecoder(self.decode) if on else None

    def decode(self, memory, mem_mask, mem_pos, frames_cls, video_cls):
        """pipeline.py:77-103 on tensors only: -> (coord [L,T,4], sted [L,1,T,2], act [L,1,T,1] | None, weights [L,1,T,T])"""
        hs, ref, time_hs, weights, _ = self.ground_decoder.run(memory, mem_mask, mem_pos, frames_cls, video_cls)  # :77-80
        coord, self.ground_decoder.last_coord = getattr(self.ground_decoder, "last_coord", None), None
        if coord is None:
            coord = ops.sigmoid(ops.add(self.bbox_embed(hs), ops.inverse_sigmoid(ref)))      # [L,T,4]  :88-93
        act = None
        if composite.ENABLED:
            sted, act = composite.time_heads(self.temp_embed, self.action_embed if self.use_actioness else None, time_hs)
            sted = sted[:, None]                                                             # [L,1,T,2]  :98
            act = act[:, None] if act is not None else None                                  # :103
        else:
            sted = self.temp_embed(time_hs)[:, None]
            if self.use_actioness:
                act = self.action_embed(time_hs)[:, None]
        return coord, sted, act, weights

    def forward(self, videos: NestedTensor, texts, logger=None) -> Dict:
        frames, frame_mask, durations = videos.decompose()
        assert len(durations) == 1, "one video per rank (datasets/build.py:150-152)"
        feat, mask, vis_pos = self.vis_encoder.forward_tokens(frames, frame_mask)            # pipeline.py:62
        n, h, w, c = feat.shape
        vis = ops.linear(feat.view(n * h * w, c), self.input_proj.weight.view(-1, c), self.input_proj.bias)  # :64
        (text_mask, text_mem, _), text_cls = self.text_encoder(texts, frames.device)         # :69
        memory, mem_mask, frames_cls, video_cls, mem_pos = self.ground_encoder.run(
            vis.view(n, h * w, -1), mask.flatten(1), vis_pos, text_mask, text_mem)           # :72-74
        # decoder + heads: tensors in, tensors out — eager launches, or ONE hipGraph launch when capture_decoder() is on
        dec = self.decode
        if self._graphed_decoder is not None and not torch.is_grad_enabled():
            dec = self._graphed_decoder
        coord, sted, act, weights = dec(memory.contiguous(), mem_mask, mem_pos, frames_cls, video_cls)
        out = {"weights": weights[-1]}                                                       # :83-85
        out["pred_boxes"] = coord[-1]
        out["pred_sted"] = sted[-1]
        if act is not None:
            out["pred_actioness"] = act[-1]
        # the per-layer tensors, still stacked [layers, ...]: VideoSTGLoss evaluates all layers in one vectorised pass
        # and would otherwise re-stack what the loop below unstacks (and pay ~70 select-backward/add launches)
        out["_stacked"] = {"pred_boxes": coord, "pred_sted": sted, "weights": weights, "pred_actioness": act}
        if self.use_aux_loss:                                                                # :106-119
            out["aux_outputs"] = []
            for i in range(coord.shape[0] - 1):
                aux = {"pred_sted": sted[i], "pred_boxes": coord[i], "weights": weights[i]}
                if act is not None:
                    aux["pred_actioness"] = act[i]
                out["aux_outputs"].append(aux)
        return out


# ------------------------------------------------------------------------------------------
# loss: models/criterion.py's API surface over ONE HIP launch for all layers and terms (csrc/loss.h)
# ------------------------------------------------------------------------------------------
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


class LossPlan:
    """Everything VideoSTGLoss derives from the targets alone (GT span, box rows, masks, Gaussian span targets,
    actioness weights), built once per batch on the host side of the input pipeline — the reference recomputes
    it inside the loss with device->host syncs (criterion.py:160-192), which stalls the launch queue."""

    def __init__(self, targets, durations, device, sigma: float, eos_coef: float):
        T = max(durations)
        b = len(durations)
        self.T, self.b = T, b
        bounds, rows = [], []
        for i, tgt in enumerate(targets):
            on = torch.where(tgt["actioness"].cpu())[0].tolist()     # (host-side annotations: no sync; the reference
            bounds.append((on[0], on[-1]))                            #  syncs on the device copy here, criterion.py:163)
            rows.extend(range(i * T + on[0], i * T + on[-1] + 1))
        self.bounds = bounds
        self.num_boxes_local = float(sum(len(t["boxs"]) for t in targets))
        self._num_boxes = None
        time_mask = torch.zeros(b, T, dtype=torch.bool)
        positive = torch.zeros(b, T, dtype=torch.bool)
        weight = torch.full((b, T), eos_coef)
        for i, d in enumerate(durations):
            time_mask[i, :d] = True
            positive[i, bounds[i][0]:bounds[i][1] + 1] = True
            weight[i, bounds[i][0]:bounds[i][1] + 1] = 1
        grid = torch.arange(T)[None, :]
        dists = []
        for idx in (0, 1):
            tgt_idx = torch.tensor([bd[idx] for bd in bounds], dtype=torch.long)
            d_ = (-((grid - tgt_idx[:, None]) ** 2) / (2 * sigma ** 2)).exp()
            dists.append(F.normalize(d_ + 1e-6, p=1, dim=1))
        pos_or_pad = positive | (~time_mask)
        # every device tensor of the plan travels in ONE asynchronous copy from one pinned staging buffer (round 4: the
        # plan is rebuilt inside every training step, as the reference rebuilds it inside its loss; ten pageable copies
        # would each stall the host until the stream drained)
        host = {
            "rows": torch.tensor(rows, dtype=torch.long),
            "tgt_boxes": torch.cat([t["boxs"].bbox.cpu() for t in targets], dim=0).float().contiguous(),
            "dist": torch.stack(dists, dim=-1).float().contiguous(),                    # [b,T,2]
            "nb_neg": ((~pos_or_pad).sum(1) + 1e-6).float(),
            "act_weight": weight.float().contiguous(),
            "actioness": torch.stack([t["actioness"].cpu() for t in targets]).float().contiguous(),
            "time_mask_u8": time_mask.to(torch.uint8),
            "pos_or_pad_u8": pos_or_pad.to(torch.uint8),
        }
        offs, total = {}, 0
        for k, v in host.items():
            offs[k] = total
            total += (v.numel() * v.element_size() + 15) & ~15
        cuda = torch.device(device).type == "cuda"
        stage = torch.empty(total, dtype=torch.uint8, pin_memory=cuda)
        for k, v in host.items():
            n = v.numel() * v.element_size()
            stage[offs[k]:offs[k] + n] = v.reshape(-1).view(torch.uint8)
        buf = stage.to(device, non_blocking=True) if cuda else stage
        self._buf = buf
        for k, v in host.items():
            n = v.numel() * v.element_size()
            setattr(self, k, buf[offs[k]:offs[k] + n].view(v.dtype).view(v.shape))
        self.row_range = (rows[0], rows[-1] + 1) if rows == list(range(rows[0], rows[-1] + 1)) else None

    # (bool / float forms of the masks, for callers that want them: views of the same buffer's contents)
    @property
    def time_mask(self):
        return self.time_mask_u8.bool()

    @property
    def time_mask_f(self):
        return self.time_mask_u8.float()

    @property
    def pos_or_pad(self):
        return self.pos_or_pad_u8.bool()

    def num_boxes(self, dev):
        """criterion.py:175-178: box count averaged over ranks, clamped to >= 1.  It depends on the targets only,
        so the 1-element all-reduce runs ONCE here, when the plan is first used — not inside every loss
        evaluation (which keeps the loss free of collectives and capturable in a hipGraph)."""
        if self._num_boxes is None:
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                nb = torch.as_tensor([self.num_boxes_local], dtype=torch.float, device=dev)
                torch.distributed.all_reduce(nb)
                self._num_boxes = torch.clamp(nb / torch.distributed.get_world_size(), min=1)  # device: no sync
            else:
                self._num_boxes = max(self.num_boxes_local, 1.0)
        return self._num_boxes


class VideoSTGLoss(nn.Module):
    """Same constructor / forward contract as models/criterion.py:11-208.  All decoder layers (main + aux) and all
    terms are evaluated by one kernel launch on the stacked [layers, ...] tensors (ops.StgLossFn), the gradient by one
    more — the reference runs ~6 x 40 small tensor ops each way."""

    def __init__(self, cfg=None, losses: Sequence[str] = ("boxes", "sted", "guided_attn", "actioness"),
                 sigma: float = 2.0, eos_coef: float = 0.3):
        super().__init__()
        self.losses = list(losses)
        self.sigma = sigma if cfg is None else cfg.SOLVER.SIGMA
        self.eos_coef = eos_coef if cfg is None else cfg.SOLVER.EOS_COEF
        self.weight_dict = None      # set (build_model does) to have the loss kernel also emit the weighted total
        self._wmat_cache = None
        self._last = None

    def plan(self, targets, durations, device) -> LossPlan:
        return LossPlan(targets, durations, device, self.sigma, self.eos_coef)

    def _wmat(self, nl: int, dev):
        """weight_dict laid out like the kernel's vec [5][nl] (row k = ops.LOSS_ROWS[k]; column nl-1 = the main output,
        column i < nl-1 = aux output i, models/__init__.py:11-27)"""
        wd = self.weight_dict
        if wd is None:
            return None
        key = (id(wd), nl, str(dev))
        if self._wmat_cache is None or self._wmat_cache[0] != key:
            m = torch.zeros(5, nl)
            keys = self._loss_keys()
            for k, name in enumerate(ops.LOSS_ROWS):
                if name not in keys:
                    continue
                for i in range(nl):
                    m[k, i] = float(wd.get(name if i == nl - 1 else f"{name}_{i}", 0.0))
            self._wmat_cache = (key, m.to(dev))
        return self._wmat_cache[1]

    def _loss_keys(self):
        keys = []
        if "boxes" in self.losses:
            keys += ["loss_bbox", "loss_giou"]
        for nm in ("sted", "guided_attn", "actioness"):
            if nm in self.losses:
                keys.append("loss_" + nm)
        return keys

    def forward(self, outputs, targets, durations, plan: Optional[LossPlan] = None):
        dev = outputs["pred_boxes"].device
        if plan is None:
            plan = targets if isinstance(targets, LossPlan) else self.plan(targets, durations, dev)
        aux = outputs.get("aux_outputs", [])
        layers = list(aux) + [outputs]                                              # main output last
        nl = len(layers)
        stk = outputs.get("_stacked")
        if stk is not None and stk["pred_boxes"].shape[0] != nl:
            stk = None                                                               # (aux losses disabled)
        _st = lambda key: stk[key] if stk is not None else torch.stack([l[key] for l in layers])  # noqa: E731
        coord = _st("pred_boxes")                                                   # [nl, rows, 4]
        act = None
        if "actioness" in self.losses:
            act = _st("pred_actioness").squeeze(-1)                                 # [nl,b,T]
        # every term of every layer, one launch (csrc/loss.h); gradient: one more
        vec, total = ops.StgLossFn.apply(coord, _st("pred_sted"), _st("weights"), act, plan, self._wmat(nl, dev))
        # criterion.py:168-171 — the reference overwrites pred_boxes with the GT-span rows
        for i, l in enumerate(layers):
            l["pred_boxes"] = coord[i, plan.row_range[0]:plan.row_range[1]] if plan.row_range else coord[i][plan.rows]
        losses = {}
        keys = self._loss_keys()
        for k, name in enumerate(ops.LOSS_ROWS):
            if name not in keys:
                continue
            losses[name] = vec[k, nl - 1]
            for i in range(nl - 1):
                losses[f"{name}_{i}"] = vec[k, i]
        self._last = (vec, total, nl)
        return losses

    def weighted_total(self, weight_dict):
        """sum_k w_k * loss_k of the last forward.  With `self.weight_dict` set (build_model does) the kernel has already
        summed it; any other weighting is applied to the per-layer vector here."""
        vec, total, nl = self._last
        if total is not None and (weight_dict is self.weight_dict or weight_dict == self.weight_dict):
            return total
        saved, self.weight_dict = self.weight_dict, weight_dict
        try:
            self._wmat_cache = None
            m = self._wmat(nl, vec.device)
        finally:
            self.weight_dict, self._wmat_cache = saved, None
        return (vec * m).sum()


class PostProcess(nn.Module):
    """models/post_processor.py:13-55; the T x T temporal map + argmax runs as one HIP kernel for the batch."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, frames_id, durations):
        sted, boxes = outputs["pred_sted"], outputs["pred_boxes"]
        assert len(boxes) == len(target_sizes)
        img_h, img_w = target_sizes.unbind(1)
        scale = torch.stack([img_w, img_h, img_w, img_h], dim=1)
        pred_boxes = (box_cxcywh_to_xyxy(boxes) * scale).clamp(min=0)
        idx = ops.temporal_map_argmax(sted, durations).cpu().tolist()
        steds = [[frames_id[b][s], frames_id[b][e] + 1] for b, (s, e) in enumerate(idx)]
        return pred_boxes, steds


def weight_dict(cfg=None, n_dec: int = 6):
    """models/__init__.py:11-27 (VidSTG yaml coefficients by default)."""
    if cfg is None:
        base = {"loss_bbox": 5.0, "loss_giou": 3.0, "loss_sted": 10.0, "loss_actioness": 2.0,
                "loss_guided_attn": 1.0}
    else:
        base = {"loss_bbox": cfg.SOLVER.BBOX_COEF, "loss_giou": cfg.SOLVER.GIOU_COEF, "loss_sted": cfg.SOLVER.TEMP_COEF}
        if cfg.MODEL.STCAT.USE_ACTION:
            base["loss_actioness"] = cfg.SOLVER.ACTIONESS_COEF
        if cfg.SOLVER.USE_ATTN:
            base["loss_guided_attn"] = cfg.SOLVER.ATTN_COEF
        n_dec = cfg.MODEL.STCAT.DEC_LAYERS
    wd = dict(base)
    for i in range(n_dec - 1):
        wd.update({f"{k}_{i}": v for k, v in base.items()})
    return wd


def build_model(cfg=None, text_encoder=None):
    model = STCATNet(cfg, text_encoder)
    criterion, wd = VideoSTGLoss(cfg), weight_dict(cfg)
    criterion.weight_dict = wd
    return model, criterion, wd


def build_postprocessors():
    return PostProcess()


@torch.no_grad()
def linear_interp(frame_ids, boxes: torch.Tensor):
    """engine/evaluate.py:11-35 on tensors: boxes [n,4] of the frames `frame_ids` (any order, no duplicates) ->
    (all ids from min to max, boxes [max-min+1, 4]) with the missing frames filled by linear interpolation between
    their two neighbours: box(left + s) = box(left) + s * (box(right) - box(left)) / (right - left).
    One gather + one fused multiply-add on the device instead of the reference's per-frame Python loop."""
    ids = torch.as_tensor(frame_ids, dtype=torch.int64)
    order = torch.argsort(ids)
    ids = ids[order]
    boxes = boxes[order.to(boxes.device)]
    n = ids.numel()
    if n < 2:
        return ids.tolist(), boxes
    full = torch.arange(int(ids[0]), int(ids[-1]) + 1, dtype=torch.int64)
    left = torch.searchsorted(ids, full, right=True) - 1                      # last given frame <= f
    right = torch.clamp(left + 1, max=n - 1)
    step = (full - ids[left]).to(torch.float64)
    interval = (ids[right] - ids[left]).clamp(min=1).to(torch.float64)
    dev = boxes.device
    bl, br = boxes[left.to(dev)].double(), boxes[right.to(dev)].double()
    out = bl + step.to(dev)[:, None] * ((br - bl) / interval.to(dev)[:, None])   # same operation order as the reference
    return full.tolist(), out


def evaluate_video(model, postprocessor, videos: NestedTensor, texts, target_sizes, frame_ids: List[List[int]],
                   interpolate: bool = False):
    """Counterpart of do_eval's per-batch body (engine/evaluate.py:97-119 with single_forward :38-77): the clip is
    split into its even and odd frames, each half goes through the model and PostProcess, boxes are merged per
    frame id and the temporal prediction is the union of the two spans."""
    durations = videos.durations
    boxes_by_frame, spans = {}, []
    for start in (0, 1):
        sub = videos.subsample(2, start)
        ids = [fid[start::2] for fid in frame_ids]
        sizes = torch.cat([ts[start::2] for ts in torch.split(target_sizes, durations)], dim=0)
        out = model(sub, texts)
        boxes, sted = postprocessor(out, sizes, ids, sub.durations)
        at = 0
        for v, fid in enumerate(ids):
            for k, f in enumerate(fid):
                boxes_by_frame[(v, f)] = boxes[at + k]
            at += len(fid)
        spans.append(sted)
    union = [[min(a[0], b[0]), max(a[1], b[1])] for a, b in zip(*spans)]
    if interpolate:  # engine/evaluate.py:114: frames the sampler skipped get interpolated boxes
        dense = {}
        for v in range(len(frame_ids)):
            fids = [f for (vv, f) in boxes_by_frame if vv == v]
            full, bx = linear_interp(fids, torch.stack([boxes_by_frame[(v, f)] for f in fids]))
            for k, f in enumerate(full):
                dense[(v, f)] = bx[k]
        boxes_by_frame = dense
    return boxes_by_frame, union
