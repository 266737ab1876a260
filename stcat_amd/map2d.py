"""2D temporal map head — counterpart of models/map2d_head.py (Gen2DMap :9-62, TempPredictionHead :65-127 with
TEMP_HEAD='conv', TempConvInteraction :228-250).  The reference never wires this head into STCATNet, the criterion or a
config node: it is provided as an OPTIONAL op, forward only, with the reference's parameter names
(`encoder.convs.{i}.{weight,bias}`, `predictor.{weight,bias}`) and pinned by goldens generated from the imported
reference (tests/golden/map2d.npz).  Not part of the videos/sec metric.

* Gen2DMap: adaptive pooling to N steps + the cascade of 39 MaxPool1d layers written on sparse diagonals = per valid
  cell (i, j) the range maximum over [i, j]: two small HIP kernels (csrc/pointwise.h) producing an NHWC map.
* TempConvInteraction: k x k convolutions (bias, ReLU) through the implicit-GEMM kernels of the backbone, each followed
  by the per-pixel mask-normalisation weight; the 1x1 predictor is the small-N linear kernel; eval applies
  sigmoid * mask2d.  The 'attn' variant (row / column attention) is not built.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops


def _sparse_cells(N: int, pooling_counts: Sequence[int]):
    mask = torch.zeros(N, N, dtype=torch.bool)
    mask[range(N), range(N)] = True
    stride, offset = 1, 0
    for c in pooling_counts:                       # map2d_head.py:20-28
        for _ in range(c):
            offset += stride
            mask[range(0, N - offset, stride), range(offset, N, stride)] = True
        stride *= 2
    return mask


def _mask2weight(mask2d: torch.Tensor, k: int, padding: int) -> torch.Tensor:
    """map2d_head.py:221-226 (a configuration constant: evaluated once on the host at construction)"""
    w = F.conv2d(mask2d[None, None].float(), torch.ones(1, 1, k, k), padding=padding)[0, 0]
    w[w > 0] = 1 / w[w > 0]
    return w


class Gen2DMap(nn.Module):
    def __init__(self, map_size: int = 128, pooling_counts: Sequence[int] = (15, 8, 8, 8)):
        super().__init__()
        self.map_size = map_size
        mask = _sparse_cells(map_size, pooling_counts)
        idx = mask.nonzero()
        self.register_buffer("mask2d", mask, persistent=False)
        self.register_buffer("cell_i", idx[:, 0].to(torch.int32).contiguous(), persistent=False)
        self.register_buffer("cell_j", idx[:, 1].to(torch.int32).contiguous(), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [b, T, D] -> map NHWC [b, N, N, D] (the reference returns the same values as [b, D, N, N])"""
        b, T, D = x.shape
        N = self.map_size
        x = x.contiguous()
        L.check_tensor(x)
        st = L.stream_of(x)
        pooled = torch.empty(b, N, D, device=x.device, dtype=torch.float32)
        L.call("stcat_map2d_pool", x.data_ptr(), pooled.data_ptr(), b, T, N, D, st)
        out = torch.zeros(b, N, N, D, device=x.device, dtype=torch.float32)
        L.call("stcat_map2d_cells", pooled.data_ptr(), self.cell_i.data_ptr(), self.cell_j.data_ptr(),
               int(self.cell_i.numel()), out.data_ptr(), b, N, D, st)
        return out


class _ConvStack(nn.Module):
    """parameter container with the reference's names: encoder.convs.{i}"""

    def __init__(self, d: int, k: int, n: int, first_padding: int):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(d, d, k, padding=first_padding)] + [nn.Conv2d(d, d, k) for _ in range(n - 1)])


class TempPredictionHead(nn.Module):
    """TempPredictionHead with TEMP_HEAD='conv' (map2d_head.py:65-127): forward(x [layers, b, T, D]) ->
    eval: sigmoid(scores) * mask2d, train: raw scores; [layers, b, N, N]."""

    def __init__(self, d_model: int = 256, map_size: int = 128, pooling_counts: Sequence[int] = (15, 8, 8, 8),
                 kernel_size: int = 9, conv_layers: int = 4):
        super().__init__()
        self.map_maker = Gen2DMap(map_size, pooling_counts)
        k, n = kernel_size, conv_layers
        pad0 = (k - 1) * n // 2
        self.encoder = _ConvStack(d_model, k, n, pad0)
        self.predictor = nn.Conv2d(d_model, 1, 1)
        ws: List[torch.Tensor] = [_mask2weight(self.map_maker.mask2d, k, pad0)]
        for _ in range(n - 1):
            ws.append(_mask2weight(ws[-1] > 0, k, 0))
        for i, w in enumerate(ws):
            self.register_buffer(f"weight{i}", w.contiguous(), persistent=False)
        self.k, self.n, self.pad0 = k, n, pad0
        for p in self.encoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)       # _reset_parameters (:100-103) runs before the predictor exists

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        nl, b, T, D = x.shape
        N = self.map_maker.map_size
        m = self.map_maker(x.reshape(nl * b, T, D))                              # NHWC [nl*b, N, N, D]
        for i, conv in enumerate(self.encoder.convs):
            w_ohwi = conv.weight.permute(0, 2, 3, 1).contiguous()
            m = ops.conv_fwd_raw(m, w_ohwi, None, conv.bias, None, 1, self.pad0 if i == 0 else 0, True)
            wgt = getattr(self, f"weight{i}")
            L.call("stcat_rowscale", m.data_ptr(), wgt.data_ptr(), m.shape[0] * m.shape[1] * m.shape[2], D, wgt.numel(),
                   L.stream_of(m))
        scores = ops.linear_fwd_raw(m.reshape(-1, D), self.predictor.weight.view(1, D).contiguous(), self.predictor.bias)
        scores = scores.view(nl, b, N, N)
        if self.training:
            return scores
        sig = ops.ew(L.EW_SIGMOID, scores.contiguous())
        return ops.ew(L.EW_MUL, sig, self.map_maker.mask2d.to(torch.float32).contiguous(), bmod=N * N)
