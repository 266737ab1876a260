"""CPU ORACLE — test infrastructure, NOT product code.

A plain-PyTorch fp32 (CPU) restatement of the STCAT hot path: ResNet-101 +
FrozenBN backbone, 2-D sine positions, input_proj, the cross-modal
spatial/temporal encoder, template generator, box decoder, time decoder,
prediction heads, VideoSTGLoss and PostProcess.  It is written as pure
functions over a flat ``{state_dict key: tensor}`` mapping (b = 1 video per
rank, as the reference enforces in datasets/build.py:150-152), so backward is
simply torch autograd over leaf weights.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``stcat_amd``) never does.

Pinning: the reference has no tests of its own, so this
restatement is pinned against outputs of the reference itself, imported in the
build container by ``tests/golden/make_golden.py`` and committed as
``tests/golden/*.npz`` (checked by ``tests/test_oracle_golden.py``).
Third-party arithmetic not under /root/reference: torchvision==0.11.0
ResNet-101 v1.5 (call site models/vision_model/backbone.py:115-119) and
torch.nn.MultiheadAttention (torch 1.10; call sites modal_encoder.py:212,236,
query_decoder.py:269,341,565-566,604,633) are restated from their published
definitions.

All file:line citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
NHEAD = 8
BLOCKS = (3, 4, 23, 3)

# --------------------------------------------------------------------------
# train mode: the reference's dropout sites
# --------------------------------------------------------------------------
# The functions below are the EVAL-mode arithmetic unless a `sites` object is installed with `dropout_sites(...)`.
# Then every nn.Dropout / attention-probability dropout of the reference path (modal_encoder.py:212-221,237-240;
# query_decoder.py:269,286-303,344,431-436,565-580,612,653-658; attention.py:381; net_utils.py:17-25 with
# pipeline.py:42,47) calls it, in the reference's execution order per section:
#     sites.elementwise(section, x, p) -> x'          nn.Dropout on a tensor
#     sites.probs(section, kind, attn, p, N, nh) -> attn'   dropout on softmax probabilities [N*nh, L, S];
#                                                            kind = "self" (L = S queries) | "q1" (one query per frame)
# section in {"enc", "box", "time", "heads"}.  torch's Philox stream cannot be reproduced by another implementation, so a
# parity test hands in the masks the HIP kernels drew (counter-based: csrc/stcat_rng.h) — "same arithmetic given the
# same mask" (tests/test_model_parity.py::_check_train_mode_against_oracle).
_SITES = None
P_LAYER = 0.1          # cfg.MODEL.STCAT.DROPOUT (experiments/*.yaml)
P_HEAD = 0.3           # pipeline.py:42,47


class dropout_sites:
    def __init__(self, sites):
        self.sites = sites

    def __enter__(self):
        global _SITES
        self.prev, _SITES = _SITES, self.sites
        return self.sites

    def __exit__(self, *exc):
        global _SITES
        _SITES = self.prev
        return False


def _drop(section: str, x: torch.Tensor, p: float) -> torch.Tensor:
    return x if _SITES is None else _SITES.elementwise(section, x, p)


def _drop_probs(section: str, kind: str, pr: torch.Tensor, p: float, N: int, nh: int) -> torch.Tensor:
    return pr if _SITES is None else _SITES.probs(section, kind, pr, p, N, nh)


# --------------------------------------------------------------------------
# backbone: torchvision ResNet-101 v1.5 with FrozenBatchNorm2d
# --------------------------------------------------------------------------
def frozen_bn(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """models/vision_model/backbone.py:56-66 — x*scale + bias with eps inside rsqrt."""
    w, b = sd[p + "weight"], sd[p + "bias"]
    rm, rv = sd[p + "running_mean"], sd[p + "running_var"]
    scale = w * (rv + 1e-5).rsqrt()
    bias = b - rm * scale
    return x * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def bottleneck(x: torch.Tensor, sd: SD, p: str, stride: int, has_ds: bool) -> torch.Tensor:
    """torchvision Bottleneck (v1.5: the stride sits on the 3x3 conv)."""
    idt = x
    o = F.relu(frozen_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
    o = F.relu(frozen_bn(F.conv2d(o, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2."))
    o = frozen_bn(F.conv2d(o, sd[p + "conv3.weight"]), sd, p + "bn3.")
    if has_ds:
        idt = frozen_bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1.")
    return F.relu(o + idt)


def backbone(sd: SD, frames: torch.Tensor, prefix: str = "vis_encoder.0.body.",
             return_stages: bool = False):
    """IntermediateLayerGetter(layer4) over resnet101 (backbone.py:86-97, 115-119).
    frames [n,3,H,W] -> layer4 [n,2048,H/32,W/32]."""
    x = F.conv2d(frames, sd[prefix + "conv1.weight"], stride=2, padding=3)
    x = F.relu(frozen_bn(x, sd, prefix + "bn1."))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    stages = []
    for li, nblk in enumerate(BLOCKS, start=1):
        for bi in range(nblk):
            stride = 2 if (bi == 0 and li > 1) else 1
            x = bottleneck(x, sd, f"{prefix}layer{li}.{bi}.", stride, bi == 0)
        stages.append(x)
    return (x, stages) if return_stages else x


def interp_mask(mask: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """backbone.py:100 — F.interpolate(mask.float(), size) nearest -> bool."""
    return F.interpolate(mask[None].float(), size=size).to(torch.bool)[0]


def pos_sine_2d(mask: torch.Tensor, num_pos_feats: int = 128, temperature: float = 10000.0) -> torch.Tensor:
    """PositionEmbeddingSine(normalize=True) — vision_model/position_encoding.py:70-94."""
    not_mask = ~mask
    y = not_mask.cumsum(1, dtype=torch.float32)
    x = not_mask.cumsum(2, dtype=torch.float32)
    scale = 2 * math.pi
    y = y / (y[:, -1:, :] + 1e-6) * scale
    x = x / (x[:, :, -1:] + 1e-6) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    px = x[:, :, :, None] / dim_t
    py = y[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------
# attention
# --------------------------------------------------------------------------
def _masked_softmax(scores: torch.Tensor, kpm: Optional[torch.Tensor], bsz: int, nh: int) -> torch.Tensor:
    if kpm is not None:
        L, S = scores.shape[-2:]
        scores = scores.view(bsz, nh, L, S).masked_fill(kpm[:, None, None, :], float("-inf")).view(bsz * nh, L, S)
    scores = scores - scores.max(dim=-1, keepdim=True)[0]  # attention.py:379-380
    return scores.softmax(dim=-1)


def torch_mha(sd: SD, p: str, q_in, k_in, v_in, kpm=None, nh: int = NHEAD, section: str = "", kind: str = "self"):
    """torch.nn.MultiheadAttention forward (packed in-proj).
    q_in [L,N,E], k_in/v_in [S,N,E]; returns (out [L,N,E], head-mean weights [N,L,S]).  Train mode: the dropout acts
    on the probabilities BEFORE both P·V and the head mean (F.multi_head_attention_forward)."""
    W, B = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    E = q_in.shape[-1]
    q = F.linear(q_in, W[:E], B[:E])
    k = F.linear(k_in, W[E:2 * E], B[E:2 * E])
    v = F.linear(v_in, W[2 * E:], B[2 * E:])
    L, N, _ = q.shape
    S = k.shape[0]
    hd = E // nh
    q = q.contiguous().view(L, N * nh, hd).transpose(0, 1) * (hd ** -0.5)
    k = k.contiguous().view(S, N * nh, hd).transpose(0, 1)
    v = v.contiguous().view(S, N * nh, hd).transpose(0, 1)
    pr = _masked_softmax(torch.bmm(q, k.transpose(1, 2)), kpm, N, nh)
    pr = _drop_probs(section, kind, pr, P_LAYER, N, nh)
    o = torch.bmm(pr, v).transpose(0, 1).contiguous().view(L, N, E)
    o = F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])
    return o, pr.view(N, nh, L, S).sum(dim=1) / nh


def dab_mha(sd: SD, p: str, q, k, v, kpm, nh: int = NHEAD, section: str = "box"):
    """DAB-DETR MultiheadAttention without q/k/v projections, vdim != embed_dim —
    grounding_model/attention.py:184-393 (scale :283-285, bmm :359, mask :369-375,
    max-subtracted softmax :379-380, PV :383, out_proj :386)."""
    L, N, E = q.shape
    S = k.shape[0]
    hd = E // nh
    vd = v.shape[-1] // nh
    qh = (q * (float(hd) ** -0.5)).contiguous().view(L, N * nh, hd).transpose(0, 1)
    kh = k.contiguous().view(S, N * nh, hd).transpose(0, 1)
    vh = v.contiguous().view(S, N * nh, vd).transpose(0, 1)
    pr = _masked_softmax(torch.bmm(qh, kh.transpose(1, 2)), kpm, N, nh)
    pr = _drop_probs(section, "q1", pr, P_LAYER, N, nh)                       # attention.py:381
    o = torch.bmm(pr, vh).transpose(0, 1).contiguous().view(L, N, vd * nh)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def layer_norm(x, sd: SD, p: str):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def lin(x, sd: SD, p: str):
    return F.linear(x, sd[p + "weight"], sd[p + "bias"])


def ffn(x, sd: SD, p: str, section: str = ""):
    """linear2(dropout(relu(linear1 x)))  (modal_encoder.py:239, query_decoder.py:435, 657)"""
    return lin(_drop(section, F.relu(lin(x, sd, p + "linear1.")), P_LAYER), sd, p + "linear2.")


def mlp(x, sd: SD, p: str, n_layers: int, dropout: float = 0.0, section: str = "heads"):
    """models/net_utils.py:7-26: with a dropout probability the MLP drops after EVERY layer, the last one included
    (:24 `i < self.num_layers` is always true) — temp_embed / action_embed (pipeline.py:42,47: p = 0.3)."""
    for i in range(n_layers):
        x = lin(x, sd, f"{p}layers.{i}.")
        if i < n_layers - 1:
            x = F.relu(x)
        if dropout:
            x = _drop(section, x, dropout)
    return x


# --------------------------------------------------------------------------
# cross-modal encoder
# --------------------------------------------------------------------------
def encoder_layer(sd: SD, p: str, src, kpm, pos):
    """TransformerEncoderLayer.forward (post-norm) — modal_encoder.py:228-242."""
    qk = src + pos
    a, _ = torch_mha(sd, p + "self_attn.", qk, qk, src, kpm, section="enc")
    src = layer_norm(src + _drop("enc", a, P_LAYER), sd, p + "norm1.")                      # dropout1 :237
    return layer_norm(src + _drop("enc", ffn(src, sd, p, "enc"), P_LAYER), sd, p + "norm2.")  # dropout2 :240


def cross_modal_encoder(sd: SD, feat, vis_mask, vis_pos, text_mask, text_mem,
                        n_layers: int = 6, p: str = "ground_encoder.encoder."):
    """CrossModalEncoder.forward + SpatialTemporalEncoder.forward for one video —
    modal_encoder.py:40-101 and :130-204.
    feat [n,256,h,w], vis_mask [n,h,w] bool, vis_pos [n,256,h,w],
    text_mask [1,L] bool, text_mem [L,1,256].
    Returns encoded_memory [HW+L,n,256], mask [n,HW+L], frames_cls [n,256], videos_cls [1,256]."""
    n, d, h, w = feat.shape
    vis_mask = vis_mask.clone()
    vis_mask[:, 0, 0] = False                                   # :46
    tokens = feat.flatten(2).permute(2, 0, 1)                   # :52
    pos = vis_pos.flatten(2).permute(2, 0, 1)
    L = text_mem.shape[0]
    txt = text_mem[:, 0:1].expand(L, n, d)                      # :70-77
    tmask = text_mask[0:1].expand(n, L)                         # :62-68
    x = torch.cat([tokens, txt], dim=0)                         # :80
    mask = torch.cat([vis_mask.flatten(1), tmask], dim=1)       # :81
    pos = torch.cat([pos, torch.zeros_like(txt)], dim=0)        # :82

    # SpatialTemporalEncoder: frame-cls token in row 0 (:145-151)
    x = torch.cat([sd[p + "frame_cls.weight"][None].expand(1, n, d), x], dim=0)
    kpm = torch.cat([torch.zeros(n, 1, dtype=torch.bool), mask], dim=1)
    pos = torch.cat([sd[p + "local_pos_embed.weight"][None].expand(1, n, d), pos], dim=0)
    video = sd[p + "video_cls.weight"][None]                    # [1,1,d]  :154
    tpos = sd[p + "time_embed.te"][: n + 1]                     # :155
    tmask_t = torch.zeros(1, n + 1, dtype=torch.bool)           # :156-159 (duration == t)
    for i in range(n_layers):
        x = encoder_layer(sd, f"{p}spatial_layers.{i}.", x, kpm, pos)                 # :163-168
        seq = torch.cat([video[0], x[0]], dim=0)[:, None, :]                           # [n+1,1,d] :170-177
        seq = encoder_layer(sd, f"{p}temporal_layers.{i}.", seq, tmask_t, tpos)       # :180-185
        video = seq[0:1].permute(1, 0, 2)                                              # :190
        x = torch.cat([seq[1:, 0, :][None], x[1:]], dim=0)                             # :195 (in place there)
    return x[1:], mask, x[0], video[:, 0, :]


# --------------------------------------------------------------------------
# decoders
# --------------------------------------------------------------------------
def gen_sineembed(pos_tensor: torch.Tensor) -> torch.Tensor:
    """models/net_utils.py:29-56 — [nq,b,4] -> [nq,b,512] ordered (y,x,w,h)."""
    scale = 2 * math.pi
    i = torch.arange(128, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(i, 2, rounding_mode="floor") / 128)

    def one(c):
        e = (pos_tensor[:, :, c] * scale)[:, :, None] / dim_t
        return torch.stack((e[:, :, 0::2].sin(), e[:, :, 1::2].cos()), dim=3).flatten(2)

    return torch.cat((one(1), one(0), one(2), one(3)), dim=2)


def inverse_sigmoid(x, eps: float = 1e-3):
    """models/net_utils.py:59-63."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def template_generator(sd: SD, frames_cls, videos_cls, p: str = "ground_decoder.template_generator."):
    """query_decoder.py:451-475 (text_cls is ignored there)."""
    content = lin(videos_cls, sd, p + "content_proj.")
    gamma = torch.tanh(lin(videos_cls[0], sd, p + "gamma_proj."))
    beta = torch.tanh(lin(videos_cls[0], sd, p + "beta_proj."))
    pos_query = lin(gamma * frames_cls + beta, sd, p + "anchor_proj.")
    temp_query = content[0][None].expand(frames_cls.shape[0], -1)
    return pos_query, temp_query


def box_decoder_layer(sd: SD, p: str, tgt, memory, mem_kpm, pos, query_pos, time_embed,
                      query_sine, first: bool):
    """TransformerDecoderLayer.forward, FROM_SCRATCH=True — query_decoder.py:310-438."""
    T = tgt.shape[0]
    q = lin(tgt, sd, p + "sa_qcontent_proj.") + lin(time_embed, sd, p + "sa_qtime_proj.") \
        + lin(query_pos, sd, p + "sa_qpos_proj.")
    k = lin(tgt, sd, p + "sa_kcontent_proj.") + lin(time_embed, sd, p + "sa_ktime_proj.") \
        + lin(query_pos, sd, p + "sa_kpos_proj.")
    v = lin(tgt, sd, p + "sa_v_proj.")
    a, w = torch_mha(sd, p + "self_attn.", q, k, v, torch.zeros(1, T, dtype=torch.bool), section="box")   # :341
    tgt = layer_norm(tgt + _drop("box", a, P_LAYER), sd, p + "norm1.")                     # dropout1 :344

    S, n, d = memory.shape
    qc = lin(tgt, sd, p + "ca_qcontent_proj.")
    kc = lin(memory, sd, p + "ca_kcontent_proj.")
    vv = lin(memory, sd, p + "ca_v_proj.")
    kp = lin(pos, sd, p + "ca_kpos_proj.")
    if first:                                                                              # :360-366
        qc = qc + lin(query_pos, sd, p + "ca_qpos_proj.")
        kc = kc + kp
    hd = d // NHEAD
    qs = lin(query_sine, sd, p + "ca_qpos_sine_proj.").view(T, 1, NHEAD, hd)
    qq = torch.cat([qc.view(T, 1, NHEAD, hd), qs], dim=3).view(T, 1, 2 * d)               # :373
    kk = torch.cat([kc.view(S, n, NHEAD, hd), kp.view(S, n, NHEAD, hd)], dim=3).view(S, n, 2 * d)  # :382
    q_cross = qq[:, 0, :][None]                                                            # [1,n,2d]  :387-398
    a = dab_mha(sd, p + "cross_attn.", q_cross, kk, vv, mem_kpm, section="box")            # :402-409
    tgt = layer_norm(tgt + _drop("box", a.view(1, T, d).transpose(0, 1), P_LAYER), sd, p + "norm3.")   # :419-432
    tgt = layer_norm(tgt + _drop("box", ffn(tgt, sd, p, "box"), P_LAYER), sd, p + "norm4.")            # :435-437
    return tgt, w


def box_decoder(sd: SD, memory, mem_kpm, pos, anchor, time_embed, n_layers: int = 6,
                p: str = "ground_decoder.decoder."):
    """TransformerDecoder.forward — query_decoder.py:169-247; bbox_embed is the
    top-level head (pipeline.py:50)."""
    T = anchor.shape[0]
    d = memory.shape[-1]
    out = torch.zeros(T, 1, d)
    inter, refs = [], [anchor]
    for i in range(n_layers):
        sine = gen_sineembed(anchor)                                       # :190
        query_pos = mlp(sine, sd, p + "ref_point_head.", 2)                # :191
        transf = 1 if i == 0 else mlp(out, sd, p + "query_scale.", 2)      # :194-197
        sine_q = sine[..., :d] * transf                                    # :200
        out, _ = box_decoder_layer(sd, f"{p}layers.{i}.", out, memory, mem_kpm, pos, query_pos,
                                   time_embed, sine_q, i == 0)
        tmp = mlp(out, sd, "bbox_embed.", 3)                               # :212
        new_anchor = (tmp + inverse_sigmoid(anchor)).sigmoid()             # :213-214
        if i != n_layers - 1:
            refs.append(new_anchor)
        anchor = new_anchor.detach()                                       # :219
        inter.append(layer_norm(out, sd, p + "norm."))                     # :221-229
    return torch.stack(inter).transpose(1, 2), torch.stack(refs).transpose(1, 2)


def time_decoder_layer(sd: SD, p: str, tgt, memory, mem_kpm, pos, query_pos, time_pos):
    """TimeDecoderLayer.forward — query_decoder.py:587-660."""
    T, _, d = tgt.shape
    qk = tgt + (query_pos + time_pos)
    a, w = torch_mha(sd, p + "self_attn.", qk, qk, tgt, torch.zeros(1, T, dtype=torch.bool), section="time")  # :604-610
    tgt = layer_norm(tgt + _drop("time", a, P_LAYER), sd, p + "norm1.")                        # :612
    q_cross = (tgt + query_pos)[:, 0, :][None]                                                # :618-634
    a, _ = torch_mha(sd, p + "cross_attn_image.", q_cross, memory + pos, memory, mem_kpm, section="time", kind="q1")  # :633-639
    tgt = layer_norm(tgt + _drop("time", a.view(1, T, d).transpose(0, 1), P_LAYER), sd, p + "norm3.")   # :653
    tgt = layer_norm(tgt + _drop("time", ffn(tgt, sd, p, "time"), P_LAYER), sd, p + "norm4.")           # :657-658
    return tgt, w


def time_decoder(sd: SD, memory, mem_kpm, pos, query_pos, time_pos, n_layers: int = 6,
                 p: str = "ground_decoder.temp_decoder."):
    """TimeDecoder.forward — query_decoder.py:494-550."""
    T = query_pos.shape[0]
    out = torch.zeros(T, 1, memory.shape[-1])
    inter, ws = [], []
    for i in range(n_layers):
        out, w = time_decoder_layer(sd, f"{p}layers.{i}.", out, memory, mem_kpm, pos, query_pos, time_pos)
        inter.append(layer_norm(out, sd, p + "norm."))
        ws.append(w)
    return torch.stack(inter).transpose(1, 2), torch.stack(ws)


def query_decoder(sd: SD, memory, mem_kpm, frames_cls, videos_cls, vis_pos, n_layers: int = 6):
    """QueryDecoder.forward for one video — query_decoder.py:83-147."""
    T = frames_cls.shape[0]
    pos_query, temp_query = template_generator(sd, frames_cls, videos_cls)
    anchor = pos_query.sigmoid()[:, None, :]                                 # :101, :118
    query_temporal = temp_query[:, None, :]                                  # :119
    time_embed = sd["ground_decoder.time_embed.te"][:T]                      # :120
    n_vis = vis_pos.shape[-2] * vis_pos.shape[-1]
    mpos = vis_pos.flatten(2).permute(2, 0, 1)                               # :121
    mpos = torch.cat([mpos, torch.zeros_like(memory[n_vis:])], dim=0)        # :122
    hs, ref = box_decoder(sd, memory, mem_kpm, mpos, anchor, time_embed, n_layers)
    time_hs, weights = time_decoder(sd, memory, mem_kpm, mpos, query_temporal, time_embed, n_layers)
    return hs, ref, time_hs, weights, pos_query


# --------------------------------------------------------------------------
# whole hot path
# --------------------------------------------------------------------------
def stcat_forward(sd: SD, frames, frame_mask, text, return_stages: bool = False):
    """STCATNet.forward minus the text encoder — models/pipeline.py:52-121.
    frames [T,3,H,W]; frame_mask [T,H,W] bool; text = ((mask[1,L], mem[L,1,256], _), cls[1,256])."""
    (text_mask, text_mem, _), _text_cls = text
    feat4, stages = backbone(sd, frames, return_stages=True)
    m = interp_mask(frame_mask, feat4.shape[-2:])
    vis_pos = pos_sine_2d(m)
    feat = F.conv2d(feat4, sd["input_proj.weight"], sd["input_proj.bias"])   # :64
    memory, mask, frames_cls, videos_cls = cross_modal_encoder(sd, feat, m, vis_pos, text_mask, text_mem)
    hs, ref, time_hs, weights, pos_query = query_decoder(sd, memory, mask, frames_cls, videos_cls, vis_pos)
    coord = (mlp(hs, sd, "bbox_embed.", 3) + inverse_sigmoid(ref)).sigmoid().flatten(1, 2)   # :88-93
    head_p = P_HEAD if _SITES is not None else 0.0
    sted = mlp(time_hs, sd, "temp_embed.", 2, dropout=head_p)                                # :98
    act = mlp(time_hs, sd, "action_embed.", 2, dropout=head_p)                               # :103
    out = {"pred_boxes": coord[-1], "pred_sted": sted[-1], "pred_actioness": act[-1], "weights": weights[-1]}
    out["aux_outputs"] = [
        {"pred_sted": sted[i], "pred_boxes": coord[i], "weights": weights[i], "pred_actioness": act[i]}
        for i in range(coord.shape[0] - 1)
    ]
    if return_stages:
        out["_stages"] = {
            "layer1": stages[0], "layer2": stages[1], "layer3": stages[2], "layer4": feat4,
            "vis_pos": vis_pos, "input_proj": feat, "encoded_memory": memory, "frames_cls": frames_cls,
            "videos_cls": videos_cls, "pos_query": pos_query, "hs": hs, "ref": ref, "time_hs": time_hs,
            "weights": weights,
        }
    return out


# --------------------------------------------------------------------------
# criterion / post-process (API surface rows 18, 20 of SURVEY.md §8a)
# --------------------------------------------------------------------------
def box_cxcywh_to_xyxy(x):
    """utils/box_utils.py:63-66."""
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def giou_diag(a, b):
    """diag(generalized_box_iou(a, b)) — utils/box_utils.py:75-113 (xyxy boxes)."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, :2], b[:, :2])
    rb = torch.min(a[:, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area_a + area_b - inter
    iou = inter / union
    lt2 = torch.min(a[:, :2], b[:, :2])
    rb2 = torch.max(a[:, 2:], b[:, 2:])
    wh2 = (rb2 - lt2).clamp(min=0)
    area = wh2[:, 0] * wh2[:, 1]
    return iou - (area - union) / area


def _sted_kl(logits, target_idx, sigma, eps=1e-6):
    """One of the two KL terms of loss_sted — models/criterion.py:74-108."""
    T = logits.shape[1]
    dist = (-((torch.arange(T)[None, :] - target_idx[:, None]) ** 2) / (2 * sigma ** 2)).exp()
    dist = F.normalize(dist + eps, p=1, dim=1)
    prob = logits.softmax(1)
    return prob * ((prob + eps) / dist).log()


def criterion_layer(out, target_boxes, actioness, s: int, e: int, num_boxes: float,
                    sigma: float, eos_coef: float):
    """The four losses for one decoder layer, one video of T frames, GT span [s, e]
    inclusive — models/criterion.py:26-130 with time_mask all True (duration == T)."""
    T = out["pred_sted"].shape[1]
    boxes = out["pred_boxes"][s:e + 1]                                            # :165-171
    losses = {}
    losses["loss_bbox"] = F.l1_loss(boxes, target_boxes, reduction="none").sum() / max(num_boxes, 1)
    losses["loss_giou"] = (1 - giou_diag(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(target_boxes))).sum() \
        / max(num_boxes, 1)
    sted = out["pred_sted"]
    ts, te = torch.tensor([s]), torch.tensor([e])
    losses["loss_sted"] = (_sted_kl(sted[:, :, 0], ts, sigma) + _sted_kl(sted[:, :, 1], te, sigma)).mean()
    w = out["weights"]                                                            # [1,T,T]  :111-130
    positive = torch.zeros(1, T, dtype=torch.bool)
    positive[0, s:e + 1] = True
    la = -(1 - w + 1e-6).log()
    la = la.masked_fill(positive[:, :, None], 0)
    nb_neg = (~positive).sum(1) + 1e-6
    losses["loss_guided_attn"] = (la.sum(2) / nb_neg[:, None]).sum(1).mean()
    pa = out["pred_actioness"].squeeze(-1)                                        # :46-62
    weight = torch.full(pa.shape, eos_coef)
    weight[0, s:e + 1] = 1
    losses["loss_actioness"] = F.binary_cross_entropy_with_logits(
        pa, actioness[None].float(), weight=weight, reduction="none").mean()
    return losses


def criterion(out, actioness, target_boxes, sigma: float = 2.0, eos_coef: float = 0.3, world: int = 1):
    """VideoSTGLoss.forward for one video — models/criterion.py:151-208."""
    idx = torch.where(actioness)[0]
    s, e = int(idx[0]), int(idx[-1])
    num_boxes = max(float(target_boxes.shape[0]) / world, 1.0)                    # :174-178
    losses = criterion_layer(out, target_boxes, actioness, s, e, num_boxes, sigma, eos_coef)
    for i, aux in enumerate(out["aux_outputs"]):
        for k, v in criterion_layer(aux, target_boxes, actioness, s, e, num_boxes, sigma, eos_coef).items():
            losses[f"{k}_{i}"] = v
    return losses


def weight_dict(bbox=5.0, giou=3.0, sted=10.0, act=2.0, attn=1.0, n_dec: int = 6):
    """models/__init__.py:11-27 with the VidSTG yaml coefficients."""
    base = {"loss_bbox": bbox, "loss_giou": giou, "loss_sted": sted, "loss_actioness": act,
            "loss_guided_attn": attn}
    wd = dict(base)
    for i in range(n_dec - 1):
        wd.update({f"{k}_{i}": v for k, v in base.items()})
    return wd


def total_loss(losses, wd=None):
    wd = wd or weight_dict()
    return sum(losses[k] * wd[k] for k in losses if k in wd)


def post_process(pred_sted, pred_boxes, sizes, frame_ids: Sequence[int], duration: int):
    """PostProcess.forward for one video — models/post_processor.py:16-55.
    Returns (boxes xyxy in pixels [T,4], [start_frame, end_frame+1], flat argmax index)."""
    boxes = box_cxcywh_to_xyxy(pred_boxes)
    img_h, img_w = sizes.unbind(1)
    boxes = (boxes * torch.stack([img_w, img_h, img_w, img_h], dim=1)).clamp(min=0)
    _, t, _ = pred_sted.shape
    neg = -1e32
    m = (torch.ones(t, t) * neg).tril(0)
    m[duration:, :] = neg
    m[:, duration:] = neg
    m = m + F.log_softmax(pred_sted[0, :, 0], dim=0)[:, None] + F.log_softmax(pred_sted[0, :, 1], dim=0)[None, :]
    flat = int(m.flatten().max(dim=0)[1])
    s, e = flat // t, flat % t
    return boxes, [frame_ids[s], frame_ids[e] + 1], flat


def linear_interp(bbox_dict):
    """engine/evaluate.py:11-35 restated (plain Python on lists, as the reference): {frame_id: [[x1,y1,x2,y2]]} ->
    the same dict with every missing frame between neighbours filled by linear interpolation."""
    fids = sorted(bbox_dict)
    if len(fids) < 2:
        return bbox_dict
    for lf, rf in zip(fids[:-1], fids[1:]):
        gap = rf - lf
        if gap > 1:
            d = [(bbox_dict[rf][0][c] - bbox_dict[lf][0][c]) / gap for c in range(4)]
            for step in range(1, gap):
                bbox_dict[lf + step] = [[bbox_dict[lf][0][c] + step * d[c] for c in range(4)]]
    fids = sorted(bbox_dict)
    assert max(fids) - min(fids) + 1 == len(fids)
    return {f: bbox_dict[f] for f in fids}
