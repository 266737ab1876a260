"""Cross-modal encoder and query decoders on the HIP ops.

Drop-ins for ``models.grounding_model.build_encoder`` / ``build_decoder`` (grounding_model/__init__.py:5-9):
identical module trees and state-dict keys, identical forward contracts
(modal_encoder.py:40-101, query_decoder.py:83-147).  Internally tokens are kept batch-first
([frames, tokens, 256] rows) — attention here is per frame (modal_encoder.py:161-168) and one query per
frame (query_decoder.py:386-417), so a frame's tokens are one contiguous row block — and converted to the
reference's [tokens, frames, 256] views only at the module boundary.  One video per rank (b == 1), as
the reference enforces (datasets/build.py:150-152); this is asserted.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import composite, ops, plans
from .misc import NestedTensor

D_MODEL = 256
NHEAD = 8


def _seq_sine_table(rows: int, d: int = D_MODEL) -> torch.Tensor:
    """SeqEmbeddingSine buffer — grounding_model/position_encoding.py:23-33."""
    position = torch.arange(rows).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2) * (-math.log(10000.0) / d))
    te = torch.zeros(rows, 1, d)
    te[:, 0, 0::2] = torch.sin(position * div_term)
    te[:, 0, 1::2] = torch.cos(position * div_term)
    return te


class SeqEmbeddingSine(nn.Module):
    def __init__(self, max_len: int, d_model: int = D_MODEL):
        super().__init__()
        self.register_buffer("te", _seq_sine_table(max_len, d_model))

    def forward(self, ln: int) -> torch.Tensor:
        return self.te[:ln]


class MLP(nn.Module):
    """models/net_utils.py:7-26.  In train mode the reference applies dropout after EVERY layer, the output layer
    included (`i < self.num_layers` is always true, net_utils.py:24) — mirrored here."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, dropout=0):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.dropout_p = dropout

    def forward(self, x):
        p = self.dropout_p if self.training else 0.0
        for i, layer in enumerate(self.layers):
            x = ops.dropout(ops.linear(x, layer.weight, layer.bias, relu=(i < self.num_layers - 1)), p)
        return x


def _lin(m: nn.Linear, x, res=None, relu=False):
    return ops.linear(x, m.weight, m.bias, res=res, relu=relu)


def _ln(m: nn.LayerNorm, x):
    return ops.layer_norm(x, m.weight, m.bias, eps=m.eps)


def _ln_res(m: nn.LayerNorm, w, b, x, res, p: float):
    """LayerNorm(res + dropout_p(x W^T + b)).  Eval (p = 0): the residual rides in the GEMM epilogue.  Train: the
    dropout and the residual add happen inside the LayerNorm kernel (no separate dropout pass)."""
    if p > 0.0:
        return ops.layer_norm(ops.linear(x, w, b), m.weight, m.bias, res=res, eps=m.eps, drop_p=p)
    return ops.layer_norm(ops.linear(x, w, b, res=res), m.weight, m.bias, eps=m.eps)


def _ffn_ln(norm: nn.LayerNorm, layer, x, p: float):
    """norm(x + dropout(linear2(dropout(relu(linear1 x)))))  (modal_encoder.py:239-241; query_decoder.py:435-437,
    657-659)"""
    h = ops.dropout(_lin(layer.linear1, x, relu=True), p)
    return _ln_res(norm, layer.linear2.weight, layer.linear2.bias, h, x, p)


def _xavier(module: nn.Module):
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


# --------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------
class TransformerEncoderLayer(nn.Module):
    """Post-norm layer of modal_encoder.py:207-242; self_attn is only the parameter container of
    nn.MultiheadAttention(256, 8) (packed in_proj_weight [768,256], out_proj)."""

    def __init__(self, d_model=D_MODEL, nhead=NHEAD, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.nhead = nhead
        self.dropout_p = dropout

    def run(self, x, pos, kpm, pos_is_const: bool):
        """x, pos: [B,S,256] batch-first; kpm [B,S] bool or None."""
        if composite.ENABLED:  # one autograd node per layer, hand-written backward (stcat_amd/composite.py)
            return composite.encoder_layer(self, x, pos, kpm, pos_is_const)
        D = x.shape[-1]
        p = self.dropout_p if self.training else 0.0
        (Wqk, Wv), (Bqk, Bv) = (ops.split_rows(self.self_attn.in_proj_weight, (2 * D, D)),
                                ops.split_rows(self.self_attn.in_proj_bias, (2 * D, D)))
        qk_in = ops.add_const(x, pos) if pos_is_const else ops.add(x, pos)      # q = k = src + pos   :234
        qk = ops.linear(qk_in, Wqk, Bqk)                                        # packed q|k projection
        v = ops.linear(x, Wv, Bv)                                               # value = src         :236
        a, _ = ops.mha_self_packed(qk, v, kpm, (D // self.nhead) ** -0.5, drop_p=p)
        x = _ln_res(self.norm1, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, a, x, p)   # :237-238
        return _ffn_ln(self.norm2, self, x, p)                                  # :239-241


class SpatialTemporalEncoder(nn.Module):
    """modal_encoder.py:104-204."""

    def __init__(self, num_layers=6, max_video_len=300, d_model=D_MODEL, nhead=NHEAD, ffn=2048, dropout=0.1):
        super().__init__()
        self.spatial_layers = nn.ModuleList(TransformerEncoderLayer(d_model, nhead, ffn, dropout)
                                            for _ in range(num_layers))
        self.temporal_layers = nn.ModuleList(TransformerEncoderLayer(d_model, nhead, ffn, dropout)
                                             for _ in range(num_layers))
        self.time_embed = SeqEmbeddingSine(max_video_len + 1, d_model)
        self.local_pos_embed = nn.Embedding(1, d_model)
        self.frame_cls = nn.Embedding(1, d_model)
        self.video_cls = nn.Embedding(1, d_model)
        self.num_layers = num_layers
        self.d_model = d_model

    def run(self, tokens, kpm, pos):
        """tokens/pos [n, S', 256] (visual + text rows), kpm [n, S'] -> (memory [n,S',256], frames_cls [n,256],
        video_cls [1,256]) for one video of n frames."""
        n, _, d = tokens.shape
        x = torch.cat([self.frame_cls.weight[None].expand(n, 1, d), tokens], dim=1)          # :145-149
        pos = torch.cat([self.local_pos_embed.weight[None].expand(n, 1, d), pos], dim=1)     # :151
        kpm = torch.cat([torch.zeros(n, 1, dtype=torch.bool, device=kpm.device), kpm], dim=1)
        video = self.video_cls.weight                                                        # [1,256]  :154
        tpos = self.time_embed(n + 1)[:, 0, :][None]                                         # [1,n+1,256] :155
        for i in range(self.num_layers):
            x = self.spatial_layers[i].run(x, pos, kpm, pos_is_const=False)                  # :163-168
            seq = torch.cat([video, x[:, 0, :]], dim=0)[None]                                # [1,n+1,256] :170-177
            seq = self.temporal_layers[i].run(seq, tpos, None, pos_is_const=True)            # :180-185
            video = seq[0, 0:1]                                                              # :190
            x = torch.cat([seq[0, 1:, None, :], x[:, 1:, :]], dim=1)                         # :195 (in place there)
        return x[:, 1:, :], x[:, 0, :], video


class CrossModalEncoder(plans.InvalidatesPlans, nn.Module):
    """modal_encoder.py:11-101."""

    def __init__(self, cfg=None):
        super().__init__()
        n_layers, max_len, d, nh, ffn, drop = 6, 300, D_MODEL, NHEAD, 2048, 0.1
        if cfg is not None:
            n_layers, max_len = cfg.MODEL.STCAT.ENC_LAYERS, cfg.INPUT.MAX_VIDEO_LEN
            d, nh, ffn, drop = (cfg.MODEL.STCAT.HIDDEN, cfg.MODEL.STCAT.HEADS, cfg.MODEL.STCAT.FFN_DIM,
                                cfg.MODEL.STCAT.DROPOUT)
            if cfg.MODEL.STCAT.USE_LEARN_TIME_EMBED:
                raise ValueError("USE_LEARN_TIME_EMBED=True is not implemented (both experiment files use sine)")
        if d != D_MODEL or nh != NHEAD:
            raise ValueError("stcat_amd kernels are built for HIDDEN=256, HEADS=8")
        self.d_model = d
        self.encoder = SpatialTemporalEncoder(n_layers, max_len, d, nh, ffn, drop)
        self.fusion = nn.Linear(d, d)  # defined and never called in the reference (modal_encoder.py:29)
        _xavier(self)

    def run(self, vis_tokens, vis_mask, vis_pos, text_mask, text_mem):
        """vis_tokens/vis_pos [n,HW,256], vis_mask [n,HW] bool, text_mask [1,L] bool, text_mem [L,1,256]."""
        n, hw, d = vis_tokens.shape
        L = text_mem.shape[0]
        if composite.ENABLED:
            # masks (bytes, 1 = padding): column 0 = the frame [CLS] slot the encoder prepends (:147), then the visual
            # tokens with token 0 forced valid (:46), then the text; built once, shared by all 24 attention layers
            full = torch.zeros(n, 1 + hw + L, dtype=torch.uint8, device=vis_tokens.device)
            full[:, 1:1 + hw] = vis_mask
            full[:, 1] = 0
            full[:, 1 + hw:] = text_mask[0:1]
            mask = full[:, 1:].contiguous()
            pos = ops._zeros(vis_tokens, n, hw + L, d)                                       # :82
            ops.ew2d(ops.L.EW_COPY, vis_pos.contiguous().view(n, hw * d), out=pos.view(n, (hw + L) * d)[:, :hw * d])
            tpos = self.encoder.time_embed(n + 1)[:, 0, :][None]                             # [1,n+1,256] :155
            memory, frames_cls, video_cls = composite.encoder(self.encoder, vis_tokens, text_mem[:, 0, :], vis_pos,
                                                              full, tpos)
            return memory, mask, frames_cls, video_cls, pos
        vis_mask = vis_mask.clone()
        vis_mask[:, 0] = False                                                               # :46
        txt = text_mem[:, 0, :][None].expand(n, L, d)                                        # :70-77
        tokens = torch.cat([vis_tokens, txt], dim=1)                                         # :80
        mask = torch.cat([vis_mask, text_mask[0:1].expand(n, L)], dim=1)                     # :81
        pos = torch.cat([vis_pos, torch.zeros(n, L, d, device=vis_pos.device)], dim=1)       # :82
        memory, frames_cls, video_cls = self.encoder.run(tokens, mask, pos)
        return memory, mask, frames_cls, video_cls, pos

    def forward(self, videos: NestedTensor = None, vis_pos=None, texts: Tuple = None) -> Dict:
        feat, vis_mask, durations = videos.decompose()
        assert len(durations) == 1, "one video per rank (datasets/build.py:150-152)"
        assert vis_pos.shape[0] == sum(durations)
        n, d, H, W = feat.shape
        text_mask, text_mem, _ = texts
        vis_mask[:, 0, 0] = False  # the reference mutates the caller's mask (modal_encoder.py:46)
        memory, mask, frames_cls, video_cls, _ = self.run(
            feat.flatten(2).transpose(1, 2), vis_mask.flatten(1), vis_pos.flatten(2).transpose(1, 2),
            text_mask, text_mem)
        return {"encoded_memory": memory.transpose(0, 1), "mask": mask.bool(), "frames_cls": frames_cls,
                "videos_cls": video_cls, "durations": durations, "fea_map_size": (H, W)}


# --------------------------------------------------------------------------------------
# decoders
# --------------------------------------------------------------------------------------
class TemplateGenerator(nn.Module):
    """query_decoder.py:441-475."""

    def __init__(self, d_model=D_MODEL, query_dim=4):
        super().__init__()
        self.content_proj = nn.Linear(d_model, d_model)
        self.gamma_proj = nn.Linear(d_model, d_model)
        self.beta_proj = nn.Linear(d_model, d_model)
        self.anchor_proj = nn.Linear(d_model, query_dim)

    def run(self, frames_cls, video_cls):
        content = _lin(self.content_proj, video_cls)                       # [1,256]
        gamma = ops.tanh(_lin(self.gamma_proj, video_cls))
        beta = ops.tanh(_lin(self.beta_proj, video_cls))
        film = ops.affine_rows(frames_cls, gamma[0], beta[0])              # gamma * frames_cls + beta  :467
        pos_query = _lin(self.anchor_proj, film)                           # [T,4]
        return pos_query, content.expand(frames_cls.shape[0], -1)


class _OutProj(nn.Module):
    """Parameter container of the DAB-style MultiheadAttention (attention.py:60-113): only out_proj exists."""

    def __init__(self, vdim=D_MODEL):
        super().__init__()
        self.out_proj = nn.Linear(vdim, vdim)
        nn.init.constant_(self.out_proj.bias, 0.0)


class TransformerDecoderLayer(nn.Module):
    """query_decoder.py:250-438 with FROM_SCRATCH=True."""

    def __init__(self, d_model=D_MODEL, nhead=NHEAD, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        for nm in ("sa_qcontent_proj", "sa_qpos_proj", "sa_qtime_proj", "sa_kcontent_proj", "sa_kpos_proj",
                   "sa_ktime_proj", "sa_v_proj"):
            setattr(self, nm, nn.Linear(d_model, d_model))
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, vdim=d_model)
        for nm in ("ca_qcontent_proj", "ca_qpos_proj", "ca_kcontent_proj", "ca_kpos_proj", "ca_qtime_proj",
                   "ca_v_proj", "ca_qpos_sine_proj"):
            setattr(self, nm, nn.Linear(d_model, d_model))
        self.cross_attn = _OutProj(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.norm4 = nn.LayerNorm(d_model)
        self.nhead = nhead
        self.dropout_p = dropout

    def run(self, tgt, kc, kpos, vv, kpm, query_pos, time_embed, query_sine, first: bool):
        """tgt/query_pos/time_embed/query_sine [T,256]; kc/kpos/vv [n,S',256] = this layer's ca_kcontent_proj(memory),
        ca_kpos_proj(pos), ca_v_proj(memory) — column blocks of the layer-batched projections (QueryDecoder.run)."""
        T, D = tgt.shape
        hd = D // self.nhead
        p = self.dropout_p if self.training else 0.0
        W, Bi = self.self_attn.in_proj_weight, self.self_attn.in_proj_bias
        q = ops.add3(_lin(self.sa_qcontent_proj, tgt), _lin(self.sa_qtime_proj, time_embed),
                     _lin(self.sa_qpos_proj, query_pos))                                     # :329-338
        k = ops.add3(_lin(self.sa_kcontent_proj, tgt), _lin(self.sa_ktime_proj, time_embed),
                     _lin(self.sa_kpos_proj, query_pos))
        v = _lin(self.sa_v_proj, tgt)
        (Wq, Wk, Wv), (Bq, Bk, Bv) = ops.split_rows(W, (D, D, D)), ops.split_rows(Bi, (D, D, D))
        qp = ops.linear(q, Wq, Bq)                                                           # nn.MHA in-proj :341
        kp_ = ops.linear(k, Wk, Bk)
        vp = ops.linear(v, Wv, Bv)
        a, _ = ops.mha_self(qp[None], kp_[None], vp[None], None, hd ** -0.5, drop_p=p)
        tgt = _ln_res(self.norm1, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, a[0], tgt, p)

        qc = _lin(self.ca_qcontent_proj, tgt)
        if first:                                                                            # :360-366
            qc = ops.add(qc, _lin(self.ca_qpos_proj, query_pos))
            kpos = kpos.contiguous()  # k1 and k2 must share one leading dimension in the kernel
            kc = ops.add(kc.contiguous(), kpos)
        qs = _lin(self.ca_qpos_sine_proj, query_sine)                                        # :369
        a = ops.attn_q1(qc, qs, kc, kpos, vv, kpm, (2 * hd) ** -0.5, drop_p=p)               # :368-409
        tgt = _ln_res(self.norm3, self.cross_attn.out_proj.weight, self.cross_attn.out_proj.bias, a, tgt, p)  # :431
        return _ffn_ln(self.norm4, self, tgt, p)                                             # :435-437


class TransformerDecoder(nn.Module):
    """query_decoder.py:150-247."""

    def __init__(self, num_layers=6, d_model=D_MODEL, nhead=NHEAD, ffn=2048, dropout=0.1, query_dim=4):
        super().__init__()
        self.layers = nn.ModuleList(TransformerDecoderLayer(d_model, nhead, ffn, dropout) for _ in range(num_layers))
        self.num_layers = num_layers
        self.norm = nn.LayerNorm(d_model)
        self.query_scale = MLP(d_model, d_model, d_model, 2)
        self.ref_point_head = MLP(query_dim // 2 * d_model, d_model, d_model, 2)
        self.bbox_embed = None  # assigned by the pipeline (pipeline.py:50)
        self.d_model = d_model
        for layer_id in range(num_layers - 1):
            self.layers[layer_id + 1].ca_qpos_proj = None                                    # :166-167

    def memory_projections(self, memory, pos):
        """ca_kcontent_proj / ca_v_proj of `memory` and ca_kpos_proj of `pos` for ALL layers as three GEMMs
        with N = layers*256 (query_decoder.py:355-358 computes them layer by layer on the same inputs)."""
        L_ = self.num_layers
        cat = lambda nm, leaf: torch.cat([getattr(getattr(l, nm), leaf) for l in self.layers], dim=0)  # noqa: E731
        kc = ops.split_cols(ops.linear(memory, cat("ca_kcontent_proj", "weight"), cat("ca_kcontent_proj", "bias")), L_)
        vv = ops.split_cols(ops.linear(memory, cat("ca_v_proj", "weight"), cat("ca_v_proj", "bias")), L_)
        kp = ops.split_cols(ops.linear(pos, cat("ca_kpos_proj", "weight"), cat("ca_kpos_proj", "bias")), L_)
        return kc, kp, vv

    def run(self, memory, kpm, pos, anchor, time_embed):
        """memory/pos [n,S',256]; anchor [T,4] (sigmoid-ed template); returns hs [L,T,256], refs [L,T,4]."""
        T = anchor.shape[0]
        if composite.ENABLED:
            return composite.box_decoder(self, memory, pos, kpm, anchor, time_embed)
        kc, kp, vv = self.memory_projections(memory, pos)
        out = torch.zeros(T, self.d_model, device=memory.device)
        inter, refs = [], [anchor]
        for i, layer in enumerate(self.layers):
            sine = ops.sine_embed(anchor)                                                    # [T,512]  :190
            query_pos = self.ref_point_head(sine)                                            # :191
            sine_q = sine[:, : self.d_model]
            if i > 0:
                sine_q = ops.mul(sine_q.contiguous(), self.query_scale(out))                 # :194-200
            out = layer.run(out, kc[i], kp[i], vv[i], kpm, query_pos, time_embed, sine_q, i == 0)
            tmp = self.bbox_embed(out)                                                       # :212
            new_anchor = ops.sigmoid(ops.add(tmp, ops.inverse_sigmoid(anchor)))              # :213-214
            if i != self.num_layers - 1:
                refs.append(new_anchor)
            anchor = new_anchor.detach()                                                     # :219
            inter.append(_ln(self.norm, out))                                                # :221-229
        return torch.stack(inter), torch.stack(refs)


class TimeDecoderLayer(nn.Module):
    """query_decoder.py:553-660."""

    def __init__(self, d_model=D_MODEL, nhead=NHEAD, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.cross_attn_image = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.norm4 = nn.LayerNorm(d_model)
        self.nhead = nhead
        self.dropout_p = dropout

    def run(self, tgt, kc, vv, kpm, query_pos, qpos_time, Wcq, Bcq):
        """kc/vv [n,S',256]: this layer's key / value in-projection of (memory + pos) / memory; Wcq/Bcq: the query
        rows of cross_attn_image's packed in-projection (split once in TimeDecoder.run)."""
        T, D = tgt.shape
        hd = D // self.nhead
        p = self.dropout_p if self.training else 0.0
        (Wqk, Wv), (Bqk, Bv) = (ops.split_rows(self.self_attn.in_proj_weight, (2 * D, D)),
                                ops.split_rows(self.self_attn.in_proj_bias, (2 * D, D)))
        qk_in = ops.add(tgt, qpos_time)                                                      # :602
        qk = ops.linear(qk_in, Wqk, Bqk)
        v = ops.linear(tgt, Wv, Bv)
        a, w = ops.mha_self_packed(qk[None], v[None], None, hd ** -0.5, need_weights=True, drop_p=p)   # :604-610
        tgt = _ln_res(self.norm1, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, a[0], tgt, p)
        qc = ops.linear(ops.add(tgt, query_pos), Wcq, Bcq)                                   # :633-634
        a = ops.attn_q1(qc, None, kc, None, vv, kpm, hd ** -0.5, drop_p=p)
        tgt = _ln_res(self.norm3, self.cross_attn_image.out_proj.weight, self.cross_attn_image.out_proj.bias,
                      a, tgt, p)                                                             # :653-654
        return _ffn_ln(self.norm4, self, tgt, p), w                                          # :657-659


class TimeDecoder(nn.Module):
    """query_decoder.py:478-550."""

    def __init__(self, num_layers=6, d_model=D_MODEL, nhead=NHEAD, ffn=2048, dropout=0.1):
        super().__init__()
        self.layers = nn.ModuleList(TimeDecoderLayer(d_model, nhead, ffn, dropout) for _ in range(num_layers))
        self.norm = nn.LayerNorm(d_model)
        self.d_model = d_model

    def run(self, memory, mem_pos, kpm, query_pos, time_pos):
        """memory, mem_pos (= memory + pos) [n,S',256].  The key/value in-projections of all layers run as two
        GEMMs with N = layers*256 (query_decoder.py:633-639 applies them per layer to the same tensors)."""
        T = query_pos.shape[0]
        D = self.d_model
        nl = len(self.layers)
        wparts = [ops.split_rows(l.cross_attn_image.in_proj_weight, (D, D, D)) for l in self.layers]
        bparts = [ops.split_rows(l.cross_attn_image.in_proj_bias, (D, D, D)) for l in self.layers]
        Wk = torch.cat([w[1] for w in wparts], dim=0)
        Bk = torch.cat([b[1] for b in bparts], dim=0)
        Wv = torch.cat([w[2] for w in wparts], dim=0)
        Bv = torch.cat([b[2] for b in bparts], dim=0)
        kc = ops.split_cols(ops.linear(mem_pos, Wk, Bk), nl)
        vv = ops.split_cols(ops.linear(memory, Wv, Bv), nl)
        out = torch.zeros(T, self.d_model, device=memory.device)
        qpos_time = ops.add_const(query_pos, time_pos)                                       # query_pos + time :602
        inter, ws = [], []
        for i, layer in enumerate(self.layers):
            out, w = layer.run(out, kc[i], vv[i], kpm, query_pos, qpos_time, wparts[i][0], bparts[i][0])
            inter.append(_ln(self.norm, out))
            ws.append(w)
        return torch.stack(inter), torch.stack(ws)


class QueryDecoder(plans.InvalidatesPlans, nn.Module):
    """query_decoder.py:13-147."""

    def __init__(self, cfg=None):
        super().__init__()
        n_layers, max_len, d, nh, ffn, drop, qd = 6, 300, D_MODEL, NHEAD, 2048, 0.1, 4
        if cfg is not None:
            n_layers, max_len = cfg.MODEL.STCAT.DEC_LAYERS, cfg.INPUT.MAX_VIDEO_LEN
            d, nh, ffn, drop, qd = (cfg.MODEL.STCAT.HIDDEN, cfg.MODEL.STCAT.HEADS, cfg.MODEL.STCAT.FFN_DIM,
                                    cfg.MODEL.STCAT.DROPOUT, cfg.MODEL.STCAT.QUERY_DIM)
            if not cfg.MODEL.STCAT.FROM_SCRATCH or cfg.MODEL.STCAT.USE_LEARN_TIME_EMBED:
                raise ValueError("only FROM_SCRATCH=True / sine time embedding (both experiment files) is implemented")
        if d != D_MODEL or nh != NHEAD or qd != 4:
            raise ValueError("stcat_amd kernels are built for HIDDEN=256, HEADS=8, QUERY_DIM=4")
        self.d_model = d
        self.template_generator = TemplateGenerator(d, qd)
        self.decoder = TransformerDecoder(n_layers, d, nh, ffn, drop, qd)
        self.temp_decoder = TimeDecoder(n_layers, d, nh, ffn, drop)
        self.time_embed = SeqEmbeddingSine(max_len + 1, d)
        _xavier(self)
        for prm in self.temp_decoder.parameters():
            prm._stcat_forked_stream = True  # run() puts this module on a second stream: see dist.GradBucketReducer

    def run(self, memory, mem_kpm, mem_pos, frames_cls, video_cls):
        """memory/mem_pos [n,S',256] batch-first; returns hs [L,T,256], ref [L,T,4], time_hs [L,T,256], weights [L,1,T,T]."""
        n, S, d = memory.shape
        T = frames_cls.shape[0]
        assert n == T
        if PREFIX_TRIGGER == "entry":
            ops.run_deferred()   # (round 6: the next clip's frozen backbone prefix starts here, under the decoders' chains)
        pos_query, temp_query = self.template_generator.run(frames_cls, video_cls)          # :97-99
        anchor = ops.sigmoid(pos_query)                                                      # :101
        time_embed = self.time_embed(T)[:, 0, :]                                             # :120
        # The box decoder and the time decoder are independent chains of ~650 tiny, latency-bound launches each
        # (6 sequential layers on [T,256] states): the time decoder runs on a second HIP stream so the two chains
        # overlap on the GPU.  Autograd replays each backward node on its forward stream, so backward overlaps too.
        # (measured and rejected, round 6: both decoder chains on streams of the device's GREATEST priority while the next
        #  clip's backbone prefix runs beside them — 105.4 / 106.0 ms per C3 step against 76.7 / 77.1; like every stream
        #  created outside the measured set they serialise with one of the step's queues: profiles/r06_prefix_pipeline.log)
        fork = ops.fork_stream(memory)
        with fork:
            if composite.ENABLED:
                time_hs, weights = composite.time_decoder(self.temp_decoder, memory, mem_pos, mem_kpm, temp_query,
                                                          time_embed)
            else:
                mem_plus_pos = ops.add_const(memory, mem_pos)                                # memory + pos  :636
                time_hs, weights = self.temp_decoder.run(memory, mem_plus_pos, mem_kpm, temp_query.contiguous(),
                                                         time_embed)
        dec_out = self.decoder.run(memory, mem_kpm, mem_pos, anchor, time_embed)
        hs, ref = dec_out[0], dec_out[1]
        self.last_coord = dec_out[2] if len(dec_out) > 2 else None    # the box head, evaluated inside the decoder node
        fork.join(time_hs, weights)
        if PREFIX_TRIGGER != "entry":
            ops.run_deferred()   # (experiment STCAT_PREFIX_TRIGGER=exit: behind the decoders' forward, under their backward)
        return hs, ref, time_hs, weights, pos_query

    def forward(self, memory_cache, vis_pos=None, text_cls=None):
        durations = memory_cache["durations"]
        assert len(durations) == 1, "one video per rank (datasets/build.py:150-152)"
        memory = memory_cache["encoded_memory"].transpose(0, 1)                              # -> [n,S',256]
        n, S, d = memory.shape
        pos = vis_pos.flatten(2).transpose(1, 2)
        pos = torch.cat([pos, torch.zeros(n, S - pos.shape[1], d, device=pos.device)], dim=1)  # :121-122
        hs, ref, time_hs, weights, _ = self.run(memory.contiguous(), memory_cache["mask"], pos.contiguous(),
                                                memory_cache["frames_cls"], memory_cache["videos_cls"])
        # reference layouts: hs/ref/time_hs [layers, b, T, *], weights [layers, b, T, T]
        return [hs[:, None], ref[:, None]], (time_hs[:, None], weights)


PREFIX_TRIGGER = __import__("os").environ.get("STCAT_PREFIX_TRIGGER", "entry")


def build_encoder(cfg=None) -> CrossModalEncoder:
    return CrossModalEncoder(cfg)


def build_decoder(cfg=None) -> QueryDecoder:
    return QueryDecoder(cfg)
