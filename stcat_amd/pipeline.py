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

from . import ops
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


class STCATNet(nn.Module):
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

    def forward(self, videos: NestedTensor, texts, logger=None) -> Dict:
        frames, frame_mask, durations = videos.decompose()
        assert len(durations) == 1, "one video per rank (datasets/build.py:150-152)"
        feat, mask, vis_pos = self.vis_encoder.forward_tokens(frames, frame_mask)            # pipeline.py:62
        n, h, w, c = feat.shape
        vis = ops.linear(feat.view(n * h * w, c), self.input_proj.weight.view(-1, c), self.input_proj.bias)  # :64
        (text_mask, text_mem, _), text_cls = self.text_encoder(texts, frames.device)         # :69
        memory, mem_mask, frames_cls, video_cls, mem_pos = self.ground_encoder.run(
            vis.view(n, h * w, -1), mask.flatten(1), vis_pos, text_mask, text_mem)           # :72-74
        hs, ref, time_hs, weights, _ = self.ground_decoder.run(memory.contiguous(), mem_mask, mem_pos,
                                                               frames_cls, video_cls)       # :77-80
        out = {"weights": weights[-1]}                                                       # :83-85
        coord = ops.sigmoid(ops.add(self.bbox_embed(hs), ops.inverse_sigmoid(ref)))          # [L,T,4]  :88-93
        out["pred_boxes"] = coord[-1]
        sted = self.temp_embed(time_hs)[:, None]                                             # [L,1,T,2]  :98
        out["pred_sted"] = sted[-1]
        act = None
        if self.use_actioness:
            act = self.action_embed(time_hs)[:, None]                                        # :103
            out["pred_actioness"] = act[-1]
        if self.use_aux_loss:                                                                # :106-119
            out["aux_outputs"] = []
            for i in range(coord.shape[0] - 1):
                aux = {"pred_sted": sted[i], "pred_boxes": coord[i], "weights": weights[i]}
                if act is not None:
                    aux["pred_actioness"] = act[i]
                out["aux_outputs"].append(aux)
        return out


# ------------------------------------------------------------------------------------------
# loss (API surface kept; stays PyTorch autograd on small tensors — SURVEY.md §2 #11)
# ------------------------------------------------------------------------------------------
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def paired_giou(a, b):
    """diag of generalized_box_iou (utils/box_utils.py:90-113) for matched xyxy boxes."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, 2:], b[:, 2:]) - torch.max(a[:, :2], b[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area_a + area_b - inter
    wh_c = (torch.max(a[:, 2:], b[:, 2:]) - torch.min(a[:, :2], b[:, :2])).clamp(min=0)
    hull = wh_c[:, 0] * wh_c[:, 1]
    return inter / union - (hull - union) / hull


class VideoSTGLoss(nn.Module):
    """Same constructor / forward contract as models/criterion.py:11-208."""

    def __init__(self, cfg=None, losses: Sequence[str] = ("boxes", "sted", "guided_attn", "actioness"),
                 sigma: float = 2.0, eos_coef: float = 0.3):
        super().__init__()
        self.losses = list(losses)
        self.sigma = sigma if cfg is None else cfg.SOLVER.SIGMA
        self.eos_coef = eos_coef if cfg is None else cfg.SOLVER.EOS_COEF

    def _layer_losses(self, out, tgt_boxes, actioness, bounds, num_boxes, time_mask, positive):
        res = {}
        dev = time_mask.device
        T = time_mask.shape[1]
        if "boxes" in self.losses:
            src = out["pred_boxes"]
            res["loss_bbox"] = F.l1_loss(src, tgt_boxes, reduction="none").sum() / max(num_boxes, 1)
            res["loss_giou"] = (1 - paired_giou(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt_boxes))).sum() \
                / max(num_boxes, 1)
        if "sted" in self.losses:
            sted = out["pred_sted"].masked_fill(~time_mask[:, :, None], -1e32)
            grid = torch.arange(T, device=dev)[None, :]
            total = 0
            for col, idx in ((0, 0), (1, 1)):
                tgt = torch.tensor([b[idx] for b in bounds], dtype=torch.long, device=dev)
                dist = F.normalize((-((grid - tgt[:, None]) ** 2) / (2 * self.sigma ** 2)).exp() + 1e-6, p=1, dim=1)
                prob = sted[:, :, col].softmax(1)
                total = total + prob * ((prob + 1e-6) / dist).log() * time_mask
            res["loss_sted"] = total.mean()
        if "guided_attn" in self.losses:
            w = out["weights"]
            pos = positive | (~time_mask)
            la = (-(1 - w + 1e-6).log()).masked_fill(pos[:, :, None], 0)
            nb_neg = (~pos).sum(1) + 1e-6
            res["loss_guided_attn"] = (la.sum(2) / nb_neg[:, None]).sum(1).mean()
        if "actioness" in self.losses:
            pa = out["pred_actioness"].squeeze(-1)
            weight = torch.full(pa.shape, self.eos_coef, device=dev)
            for i, (s, e) in enumerate(bounds):
                weight[i, s:e + 1] = 1
            la = F.binary_cross_entropy_with_logits(pa, actioness, weight=weight, reduction="none")
            res["loss_actioness"] = (la * time_mask).mean()
        return res

    def forward(self, outputs, targets, durations):
        T = max(durations)
        dev = outputs["pred_boxes"].device
        bounds, rows = [], []
        for i, tgt in enumerate(targets):
            on = torch.where(tgt["actioness"])[0].tolist()
            bounds.append((on[0], on[-1]))
            rows.extend(range(i * T + on[0], i * T + on[-1] + 1))
        rows = torch.tensor(rows, dtype=torch.long, device=dev)
        outputs["pred_boxes"] = outputs["pred_boxes"][rows]                                  # criterion.py:168
        for aux in outputs.get("aux_outputs", []):
            aux["pred_boxes"] = aux["pred_boxes"][rows]
        num_boxes = torch.as_tensor([float(sum(len(t["boxs"]) for t in targets))], device=dev)
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(num_boxes)                                          # criterion.py:175-177
            world = torch.distributed.get_world_size()
        num_boxes = torch.clamp(num_boxes / world, min=1).item()
        time_mask = torch.zeros(len(durations), T, dtype=torch.bool, device=dev)
        positive = torch.zeros_like(time_mask)
        for i, d in enumerate(durations):
            time_mask[i, :d] = True
            positive[i, bounds[i][0]:bounds[i][1] + 1] = True
        tgt_boxes = torch.cat([t["boxs"].bbox for t in targets], dim=0).to(dev)
        actioness = torch.stack([t["actioness"] for t in targets]).float().to(dev)
        losses = self._layer_losses(outputs, tgt_boxes, actioness, bounds, num_boxes, time_mask, positive)
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            for k, v in self._layer_losses(aux, tgt_boxes, actioness, bounds, num_boxes, time_mask, positive).items():
                losses[f"{k}_{i}"] = v
        return losses


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
    return model, VideoSTGLoss(cfg), weight_dict(cfg)


def build_postprocessors():
    return PostProcess()
