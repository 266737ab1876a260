"""Boundary types of the hot path (counterparts of the reference's utils/misc.py:41-97 and
utils/bounding_box.py BoxList.bbox) — plain containers, no arithmetic."""
from __future__ import annotations

from typing import List, Sequence

import torch


class NestedTensor:
    """frames [sum(T),C,H,W] + pad mask [sum(T),H,W] (True = pad) + per-video durations."""

    def __init__(self, tensors: torch.Tensor, mask: torch.Tensor, durations: Sequence[int]):
        self.tensors = tensors
        self.mask = mask
        self.durations = list(durations)

    def to(self, *args, **kwargs) -> "NestedTensor":
        mask = self.mask.to(*args, **kwargs) if self.mask is not None else None
        return type(self)(self.tensors.to(*args, **kwargs), mask, self.durations)

    def decompose(self):
        return self.tensors, self.mask, self.durations

    def subsample(self, stride: int, start_idx: int = 0) -> "NestedTensor":
        """every stride-th frame of each video (engine/evaluate.py:97-104 two-pass eval)."""
        frames, masks, durs = [], [], []
        offset = 0
        for d in self.durations:
            sel = slice(offset + start_idx, offset + d, stride)
            frames.append(self.tensors[sel])
            masks.append(self.mask[sel])
            durs.append(frames[-1].shape[0])
            offset += d
        return NestedTensor(torch.cat(frames, dim=0), torch.cat(masks, dim=0), durs)

    @classmethod
    def from_tensor_list(cls, clips: List[torch.Tensor]) -> "NestedTensor":
        assert clips[0].ndim == 4
        c = max(x.shape[1] for x in clips)
        h = max(x.shape[2] for x in clips)
        w = max(x.shape[3] for x in clips)
        durs = [x.shape[0] for x in clips]
        out = torch.zeros((sum(durs), c, h, w), dtype=clips[0].dtype, device=clips[0].device)
        mask = torch.ones((sum(durs), h, w), dtype=torch.bool, device=clips[0].device)
        at = 0
        for x in clips:
            t, cc, hh, ww = x.shape
            out[at:at + t, :cc, :hh, :ww].copy_(x)
            mask[at:at + t, :hh, :ww] = False
            at += t
        return cls(out, mask, durs)

    def __repr__(self):
        return f"NestedTensor({tuple(self.tensors.shape)}, durations={self.durations})"


class BoxList:
    """Minimal target-box holder: `.bbox` [N,4] float32 and len()."""

    def __init__(self, bbox, image_size=None, mode: str = "xyxy"):
        bbox = torch.as_tensor(bbox, dtype=torch.float32)
        if bbox.ndim != 2 or bbox.shape[-1] != 4:
            raise ValueError(f"bbox should be [N,4], got {tuple(bbox.shape)}")
        self.bbox = bbox
        self.size = image_size
        self.mode = mode

    def to(self, device):
        return BoxList(self.bbox.to(device), self.size, self.mode)

    def __len__(self):
        return self.bbox.shape[0]
