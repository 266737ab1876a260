"""2D temporal map head — counterpart of models/map2d_head.py: Gen2DMap (:9-62), TempPredictionHead (:65-127) with both
interaction variants, TempConvInteraction (TEMP_HEAD='conv', :228-250) and the row / column attention encoder
(TEMP_HEAD='attn', :130-205).  The reference never wires this head into STCATNet, the criterion or a config node
(SURVEY.md §2 #10): it is provided as an OPTIONAL op with the reference's parameter names
(`encoder.convs.{i}.{weight,bias}` / `encoder.layers.{i}.self_attn_row.in_proj_weight` ..., `predictor.{weight,bias}`),
forward AND backward (train mode returns the raw scores a loss would consume, :122-124), pinned by goldens generated
from the imported reference (tests/golden/map2d.npz).  Not part of the videos/sec metric.

* Gen2DMap: adaptive pooling to N steps + the cascade of 39 MaxPool1d layers written on sparse diagonals = per valid
  cell (i, j) the range maximum over [i, j]: two small HIP kernels (csrc/pointwise.h) producing an NHWC map; backward =
  the gradient of a cell to the first maximum of its range, then through the adaptive pooling.
* TempConvInteraction: k x k convolutions (bias, ReLU) through the implicit-GEMM kernels of the backbone (forward, data
  gradient, weight gradient), each followed by the per-pixel mask-normalisation weight; the 1x1 predictor is the
  small-N linear kernel; eval applies sigmoid * mask2d.
* 'attn': per map, `TEMP_PRED_LAYERS` post-norm layers of attention along the rows (batched over the columns), then
  along the columns (batched over the rows), FFN.  The reference passes `mask2d` (True = VALID cell) as
  `key_padding_mask` (True = ignore, :171-183): the valid cells are the ones masked out.  Mirrored as is.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

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


class _Gen2DMapFn(Function):
    """x [b,T,D] -> NHWC map [b,N,N,D] (map2d_head.py:38-62)"""

    @staticmethod
    def forward(ctx, x, cell_i, cell_j, N):
        b, T, D = x.shape
        x = x.contiguous()
        L.check_tensor(x)
        st = L.stream_of(x)
        pooled = torch.empty(b, N, D, device=x.device, dtype=torch.float32)
        L.call("stcat_map2d_pool", x.data_ptr(), pooled.data_ptr(), b, T, N, D, st)
        out = torch.zeros(b, N, N, D, device=x.device, dtype=torch.float32)
        L.call("stcat_map2d_cells", pooled.data_ptr(), cell_i.data_ptr(), cell_j.data_ptr(), int(cell_i.numel()),
               out.data_ptr(), b, N, D, st)
        ctx.save_for_backward(x, pooled, cell_i, cell_j)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, dmap):
        x, pooled, cell_i, cell_j = ctx.saved_tensors
        b, T, D = x.shape
        N = ctx.N
        dmap = dmap.contiguous()
        st = L.stream_of(dmap)
        dpooled = torch.zeros_like(pooled)
        L.call("stcat_map2d_cells_bwd", pooled.data_ptr(), cell_i.data_ptr(), cell_j.data_ptr(), int(cell_i.numel()),
               dmap.data_ptr(), dpooled.data_ptr(), b, N, D, st)
        dx = torch.zeros_like(x)
        L.call("stcat_map2d_pool_bwd", x.data_ptr(), dpooled.data_ptr(), dx.data_ptr(), b, T, N, D, st)
        return dx, None, None, None


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
        return _Gen2DMapFn.apply(x, self.cell_i, self.cell_j, self.map_size)


class _ConvReluScaleFn(Function):
    """y = relu(conv(x, w) + bias) * pixel_weight  — one stage of TempConvInteraction (map2d_head.py:245-249), NHWC"""

    @staticmethod
    def forward(ctx, x, w, bias, pix_w, pad):
        w_ohwi = w.permute(0, 2, 3, 1).contiguous()
        y = ops.conv_fwd_raw(x.contiguous(), w_ohwi, None, bias, None, 1, pad, True)
        L.call("stcat_rowscale", y.data_ptr(), pix_w.data_ptr(), y.shape[0] * y.shape[1] * y.shape[2], y.shape[3],
               pix_w.numel(), L.stream_of(y))
        ctx.save_for_backward(x, w_ohwi, y, pix_w)
        ctx.pad = pad
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_ohwi, y, pix_w = ctx.saved_tensors
        g = dy.contiguous().clone()
        L.call("stcat_rowscale", g.data_ptr(), pix_w.data_ptr(), g.shape[0] * g.shape[1] * g.shape[2], g.shape[3],
               pix_w.numel(), L.stream_of(g))
        # pixel weights are >= 0 and the gradient carries the same factor: y > 0 <=> relu(...) > 0 wherever it matters
        g, _ = ops.act_bwd_raw(g, y, None, want_g=True, relu=True)
        dx = ops.conv_dgrad_raw(g, w_ohwi, x.shape, 1, ctx.pad) if ctx.needs_input_grad[0] else None
        dw = ops.conv_wgrad_raw(g, x, w_ohwi.shape, 1, ctx.pad).permute(0, 3, 1, 2) if ctx.needs_input_grad[1] else None
        db = ops.colsum(g.view(-1, g.shape[-1])) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None


class _ConvStack(nn.Module):
    """parameter container with the reference's names: encoder.convs.{i}"""

    def __init__(self, d: int, k: int, n: int, first_padding: int):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(d, d, k, padding=first_padding)] + [nn.Conv2d(d, d, k) for _ in range(n - 1)])


class _RowColLayer(nn.Module):
    """TransformerEncoderLayer of map2d_head.py:147-205 (parameter names as there)"""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int, dropout: float):
        super().__init__()
        self.self_attn_row = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.self_attn_col = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.nhead = nhead
        self.dropout_p = dropout

    @staticmethod
    def _attend(att: nn.MultiheadAttention, tokens: torch.Tensor, kpm: torch.Tensor, nhead: int, p: float) -> torch.Tensor:
        """nn.MultiheadAttention(q = k = v = tokens) on batch-first tokens [B,S,d]: packed in-projection, attention core,
        out-projection"""
        B, S, d = tokens.shape
        qkv = ops.linear(tokens, att.in_proj_weight, att.in_proj_bias)                         # [B,S,3d]
        a, _ = ops.mha_self(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], kpm, (d // nhead) ** -0.5, drop_p=p)
        return ops.linear(a, att.out_proj.weight, att.out_proj.bias)

    def run(self, m: torch.Tensor, mask2d: torch.Tensor) -> torch.Tensor:
        """m: ONE map, NHWC [N(i), N(j), d] (the reference permutes its [d,N,N] map to exactly this, :164)"""
        p = self.dropout_p if self.training else 0.0
        # row attention (:168-172): sequence = rows i, batch = columns j, key_padding_mask[j, i] = mask2d[j, i]
        t = m.transpose(0, 1).contiguous()                                                     # [j, i, d]
        r = self._attend(self.self_attn_row, t, mask2d, self.nhead, p).transpose(0, 1).contiguous()   # back to [i, j, d]
        # column attention (:175-182): sequence = columns j, batch = rows i, key_padding_mask[i, j] = mask2d[j, i]
        c = self._attend(self.self_attn_col, r, mask2d.t().contiguous(), self.nhead, p)
        x = ops.layer_norm(c, self.norm1.weight, self.norm1.bias, res=m, eps=self.norm1.eps, drop_p=p)      # :186-187
        h = ops.dropout(ops.linear(x, self.linear1.weight, self.linear1.bias, relu=True), p)
        h = ops.linear(h, self.linear2.weight, self.linear2.bias)
        return ops.layer_norm(h, self.norm2.weight, self.norm2.bias, res=x, eps=self.norm2.eps, drop_p=p)   # :188-190


class _RowColEncoder(nn.Module):
    """TransformerEncoder of map2d_head.py:130-145 (norm=None)"""

    def __init__(self, d_model, nhead, dim_feedforward, dropout, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(_RowColLayer(d_model, nhead, dim_feedforward, dropout) for _ in range(num_layers))


class TempPredictionHead(nn.Module):
    """TempPredictionHead (map2d_head.py:65-127): forward(x [layers, b, T, D]) -> [layers, b, N, N];
    eval: sigmoid(scores) * mask2d, train: raw scores (differentiable)."""

    def __init__(self, d_model: int = 256, map_size: int = 128, pooling_counts: Sequence[int] = (15, 8, 8, 8),
                 kernel_size: int = 9, conv_layers: int = 4, temp_head: str = "conv", nhead: int = 8,
                 dim_feedforward: int = 2048, dropout: float = 0.1, attn_layers: int = 2):
        super().__init__()
        self.map_maker = Gen2DMap(map_size, pooling_counts)
        self.temp_head = temp_head
        if temp_head == "attn":
            if d_model != nhead * 32:
                raise ValueError("the attention kernels are built for head dimension 32")
            self.encoder = _RowColEncoder(d_model, nhead, dim_feedforward, dropout, attn_layers)
        else:
            k, n = kernel_size, conv_layers
            pad0 = (k - 1) * n // 2
            self.encoder = _ConvStack(d_model, k, n, pad0)
            ws: List[torch.Tensor] = [_mask2weight(self.map_maker.mask2d, k, pad0)]
            for _ in range(n - 1):
                ws.append(_mask2weight(ws[-1] > 0, k, 0))
            for i, w in enumerate(ws):
                self.register_buffer(f"weight{i}", w.contiguous(), persistent=False)
            self.k, self.n, self.pad0 = k, n, pad0
        for p in self.encoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)       # _reset_parameters (:100-103) runs before the predictor exists
        self.predictor = nn.Conv2d(d_model, 1, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        nl, b, T, D = x.shape
        N = self.map_maker.map_size
        m = self.map_maker(x.reshape(nl * b, T, D))                              # NHWC [nl*b, N, N, D]
        if self.temp_head == "attn":
            maps = []
            for i in range(nl * b):                                              # :113-115: map by map
                mi = m[i]
                for layer in self.encoder.layers:
                    mi = layer.run(mi, self.map_maker.mask2d)
                maps.append(mi)
            m = torch.stack(maps)
        else:
            for i, conv in enumerate(self.encoder.convs):
                m = _ConvReluScaleFn.apply(m, conv.weight, conv.bias, getattr(self, f"weight{i}"),
                                           self.pad0 if i == 0 else 0)
        scores = ops.linear(m.reshape(-1, D), self.predictor.weight.view(1, D), self.predictor.bias)
        scores = scores.view(nl, b, N, N)
        if self.training:
            return scores
        with torch.no_grad():
            sig = ops.ew(L.EW_SIGMOID, scores.contiguous())
            return ops.ew(L.EW_MUL, sig, self.map_maker.mask2d.to(torch.float32).contiguous(), bmod=N * N)
