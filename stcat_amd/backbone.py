"""ResNet-101 + FrozenBatchNorm2d visual encoder on the HIP implicit-GEMM kernels.

Drop-in for the reference factory ``models.vision_model.build_vis_encoder`` (vision_model/__init__.py:5-24):
same module tree / state-dict keys (``0.body.conv1.weight`` ... ``0.body.layer4.2.bn3.running_var``), same
``forward(NestedTensor) -> (NestedTensor, pos)`` contract (backbone.py:93-102, 151-159), ``num_channels``.
The arithmetic — torchvision's ResNet-101 v1.5 topology, not part of the reference repository — runs as
one autograd function over NHWC activations: conv + FrozenBN + residual + ReLU are one kernel each,
weights live as [O,I,KH,KW] parameters in channels_last memory (physically OHWI, what the kernels read).
Stem and layer1 are frozen as in BackboneBase (backbone.py:78-85): they have no backward at all.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops, plans
from .misc import NestedTensor

import os

BLOCKS = (3, 4, 23, 3)
# the plane-format forward as N frame-range chains on N streams (default 2; 1 = off)
FORWARD_CHAINS = 1 if os.environ.get("STCAT_NO_FORWARD_CHAINS") else int(os.environ.get("STCAT_FORWARD_CHAINS", "2"))
# the same for the data-gradient chain of the backward pass: measured neutral next to the weight-gradient stream
# (86.9 vs 86.9 ms per C3 step), so opt-in
BACKWARD_CHAINS = bool(os.environ.get("STCAT_BACKWARD_CHAINS"))
COARSE_ADD = not os.environ.get("STCAT_NO_COARSE_ADD")   # downsample-branch gradient added from its own grid (round 5)
# round 6: the next clip's frozen prefix (stem + max-pool + layer1) under the current step's grounding section
PREFIX_PIPELINE = not os.environ.get("STCAT_NO_PREFIX_PIPELINE")
PREFIX_STREAM = int(os.environ.get("STCAT_PREFIX_STREAM", "2"))     # index into ops.side_stream (2 = the spare queue)
PREFIX_PRIO = int(os.environ.get("STCAT_PREFIX_PRIO", "0"))         # 0 = a measured-concurrent default-priority side stream; 1 = least priority
PREFIX_CUS = int(os.environ.get("STCAT_PREFIX_CUS", "0"))           # > 0: the prefix stream is masked to this many CUs
# the prefix blocks' convs in pieces of this many frames (0 = the forward chains' own ranges).  Default "auto": 4 frames on a
# single GPU, 0 under a live gradient exchange — measured both ways on one box (profiles/r06_prefix_pipeline.log): pieces of
# 4 gain 0.65 ms per step without a process group and LOSE 0.8 ms with one (the longer-lived prefix then also shares the
# chip with the reducer's copies and collectives)
_PREFIX_RANGE_ENV = os.environ.get("STCAT_PREFIX_RANGE", "auto")


def _prefix_range() -> int:
    if _PREFIX_RANGE_ENV != "auto":
        return int(_PREFIX_RANGE_ENV)
    sink = ops.GRAD_SINK
    return 0 if (sink is not None and getattr(sink, "comm", False)) else 4
PREFIX_EAGER = bool(os.environ.get("STCAT_PREFIX_EAGER"))           # the staged prefix launch by launch (no launch plan)
PREFIX_AT = os.environ.get("STCAT_PREFIX_AT", "decoder")             # "decoder": queued at the query decoder's entry; "backbone"


def _chain_streams(dev, k):
    """k - 1 extra streams for the forward chains (the first chain runs on the caller's stream).  They come from the
    package's one pool of side streams (ops.side_stream): the forward chains, the forked decoder and the input
    prefetcher are never busy at the same time, and HIP maps ALL streams of a process onto GPU_MAX_HW_QUEUES = 4
    hardware queues — a fifth stream shares a queue with another one and the two serialise (measured: with a live RCCL
    group, whose stream is the fifth, the second forward chain landed on the main stream's queue and the forward lost
    its overlap, +3.5 ms per step; profiles/r03_hw_queues.log)."""
    return [ops.side_stream(dev, i) for i in range(k - 1)]
PLANES = (64, 128, 256, 512)


class FrozenBatchNorm2d(nn.Module):
    """Buffers only (weight, bias, running_mean, running_var): backbone.py:16-66.  The affine form
    x*scale + bias is applied inside the conv epilogue; ``folded()`` caches scale/bias per buffer version."""

    def __init__(self, n: int):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self._cache = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)  # backbone.py:42-44
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def folded(self):
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((b.data_ptr(), b._version) for b in bufs)
        if self._cache is None or self._cache[0] != key:
            if self._cache is not None:
                # a RE-fold (load_state_dict / .to() after the first step): launch plans baked the old fold's address
                from . import plans
                plans.invalidate()
            self._cache = (key, ops.frozen_bn_fold(*bufs, eps=1e-5))
        return self._cache[1]


def _conv(cin, cout, k, stride=1, pad=0):
    m = nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)  # parameter container only
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride


class ResNet101Body(nn.Module):
    """Children named as torchvision's resnet101 up to layer4 (IntermediateLayerGetter, backbone.py:90)."""

    def __init__(self):
        super().__init__()
        self._wt_cache = ops.WeightTransposer()  # transposed conv weights for the data-gradient GEMMs (backward)
        self._wpl_cache = ops.WeightPlanes()     # bf16 hi/lo planes of every conv weight (mma mode "bf16x3p")
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        inplanes = 64
        for li, (nblk, planes) in enumerate(zip(BLOCKS, PLANES), start=1):
            blocks = []
            for bi in range(nblk):
                stride = 2 if (bi == 0 and li > 1) else 1
                ds = None
                if bi == 0:
                    ds = nn.Sequential(_conv(inplanes, planes * 4, 1, stride), FrozenBatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride, ds))
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))

    def blocks(self):
        for li in range(1, 5):
            for blk in getattr(self, f"layer{li}"):
                yield li, blk


def _ohwi(w: torch.Tensor) -> torch.Tensor:
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _stem(frames, body, s, b):
    """fp32 [n,3,H,W] (the reference's normalised NestedTensor) or uint8 [n,H,W,3] straight from the decoder: the
    uint8 form folds ToTensor + Normalize into the stem's gather."""
    if frames.dtype == torch.uint8:
        return ops.stem_u8_fwd_raw(frames.contiguous(), body.conv1.weight.contiguous(), s, b)
    return ops.stem_fwd_raw(frames.contiguous(), body.conv1.weight.contiguous(), s, b)


_PREFIX_LANES = {}


def _prefix_lane(dev):
    """the stream the next clip's prefix runs on.  Default: the spare one of the package's measured-concurrent side streams
    (ops.side_stream(dev, PREFIX_STREAM): main, forward chain / time decoder, weight gradients and this one fill the four
    hardware queues HIP gives a process).  Measured and rejected (profiles/r06_prefix_pipeline.log, same box, ms per C3
    step; off = 77.3): a stream of the device's least priority (STCAT_PREFIX_PRIO=1) 94.8, a stream masked to 192 / 128 /
    64 compute units (STCAT_PREFIX_CUS=n) 111-127 — a stream created beside torch's pool lands on a hardware queue that
    one of the step's streams already uses and the two serialise (the node timeline shows the decoders waiting for the
    whole prefix)."""
    key = str(dev)
    st = _PREFIX_LANES.get(key)
    if st is None:
        if PREFIX_CUS > 0 or PREFIX_PRIO != 0:
            st = _probed_lane(dev)
        if st is None:
            st = ops.side_stream(dev, PREFIX_STREAM)
        _PREFIX_LANES[key] = st
    return st


def _probed_lane(dev):
    """A stream made by stcat_stream_create (CU mask / priority) that does NOT share a hardware queue with the main stream,
    the forward-chain / time-decoder stream or the weight-gradient stream.  HIP hands new streams to its four hardware
    queues round-robin, and the step's four streams occupy all of them: a fifth stream serialises with whichever one it
    lands on (measured: 94-127 ms per step).  So several candidates are created and each is timed against the three busy
    streams with the 500 us spin kernel, as ops._pick_streams does; the first one that runs beside all three is kept — it
    shares its queue with the spare side stream, which the prefix then no longer uses.  None if no candidate qualifies."""
    import ctypes
    import time
    lib = ops.L.load()
    busy = [torch.cuda.current_stream(dev), ops.side_stream(dev, 0), ops.side_stream(dev, 1)]

    def run_ms(streams):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for s_ in streams:
            if lib.stcat_spin(500, s_.cuda_stream) != 0:
                raise ops.L.StcatHipError("stcat_spin failed")
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3

    single = min(run_ms(busy[:1]) for _ in range(3))
    report = []
    for _ in range(8):
        out = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = lib.stcat_stream_create(PREFIX_PRIO, PREFIX_CUS, ctypes.byref(out))
        if rc != 0 or not out.value:
            raise ops.L.StcatHipError("stcat_stream_create failed: " + lib.stcat_last_error().decode())
        cand = torch.cuda.ExternalStream(out.value, device=dev)
        run_ms([cand])
        pair = [min(run_ms([b, cand]), run_ms([b, cand])) for b in busy]
        report.append([round(x, 3) for x in pair])
        if all(x < 1.5 * single for x in pair):
            PREFIX_LANE_REPORT[str(dev)] = {"spin_ms": round(single, 3), "pairs_ms": report, "picked": len(report) - 1}
            return cand
        torch.cuda.synchronize(dev)
        lib.stcat_stream_destroy(out)
    PREFIX_LANE_REPORT[str(dev)] = {"spin_ms": round(single, 3), "pairs_ms": report, "picked": None}
    return None


PREFIX_LANE_REPORT = {}


def _prefix_blocks(body):
    """the frozen prefix of the network: stem + max-pool + the leading non-trainable bottlenecks of layer1
    (BackboneBase freezes everything below layer2, backbone.py:78-85) — no backward, a pure function of the frames"""
    out = []
    for li, blk in body.blocks():
        if li != 1 or blk.conv1.weight.requires_grad:
            break
        out.append(blk)
    return out


def _chain_cuts(n_all: int, cuda: bool):
    """the frame ranges of the forward chains (one range = no chains)"""
    if FORWARD_CHAINS > 1 and n_all >= 4 * FORWARD_CHAINS and cuda and ops.FORK_ENABLED:
        k = FORWARD_CHAINS
        cuts = [round(i * n_all / k) for i in range(k + 1)]
        return list(zip(cuts[:-1], cuts[1:]))
    return [(0, n_all)]


def _prefix_ranges(chains):
    """the frame ranges the PREFIX blocks' convs are issued on: each forward chain's range cut into pieces of at most
    _prefix_range() frames (0 = the chains' own ranges).  A launch over few frames has fewer workgroups than the chip has CUs
    (4 frames of layer1 at 448 x 448 are 196 tiles of 256 rows), and the launches of one stream run one after the other:
    the background prefix then never holds more than ~3/4 of the chip, and the decoders' dependent small launches beside it
    — whose workgroups need a CU that a plane-GEMM workgroup has vacated entirely (it owns the CU's registers and LDS) —
    always find free ones.  The in-step path uses the same pieces (bit-identical results: the tile / kernel picker depends
    on a launch's rows)."""
    step = _prefix_range()
    if step <= 0:
        return [(a, b, ci) for ci, (a, b) in enumerate(chains)]
    out = []
    for ci, (a, b) in enumerate(chains):
        for lo in range(a, b, step):
            out.append((lo, min(b, lo + step), ci))
    return out


def _prefix_forward(frames, body, out: "ops.Planes"):
    """stem + max-pool + the prefix blocks of ONE whole clip on the current stream (plane mode), the last block's output
    (+ its ReLU bit mask when `out` carries one) written into `out`.  Every conv is issued once per frame range of the
    in-node path's forward chains (_chain_cuts), one range after the other on this stream: the SAME launches with the same
    arguments as the in-node path (the tile / kernel picker depends on the rows of a launch, and another kernel variant
    sums in another order), so the result is bit-identical to it (tests/test_plans.py: *_backbone_bit_exact)."""
    blks = _prefix_blocks(body)
    wp = body._wpl_cache.fwd
    s, b = body.bn1.folded()
    x = _stem(frames, body, s, b)
    x = ops.pl_maxpool_raw(x)
    chains = [(a, b) for a, b, _ in _prefix_ranges(_chain_cuts(x.shape[0], x.t.is_cuda))]

    def conv(xin, w_, s_, b_, res, stride, pad, relu, out_=None):
        if len(chains) == 1:
            return ops.pl_conv_fwd_raw(xin, w_, s_, b_, res, stride, pad, relu, out=(out_, None) if out_ is not None else None)[0]
        n, H, W, _ = xin.shape
        Cout, KH = w_.shape[0], w_.shape[1]
        OH, OW = ops.conv_out_hw(H, W, KH, stride, pad)
        yp = out_ if out_ is not None else ops.Planes.empty(xin.t, n, OH, OW, Cout)
        for a, b2 in chains:
            ops.pl_conv_fwd_raw(xin.frames(a, b2), w_, s_, b_, res.frames(a, b2) if res is not None else None, stride, pad,
                                relu, out=(yp.frames(a, b2), None))
        return yp

    for bi, blk in enumerate(blks):
        last = bi == len(blks) - 1
        w1, w2, w3 = _ohwi(blk.conv1.weight), _ohwi(blk.conv2.weight), _ohwi(blk.conv3.weight)
        s1, b1 = blk.bn1.folded()
        s2, b2 = blk.bn2.folded()
        s3, b3 = blk.bn3.folded()
        o1 = conv(x, wp[w1.data_ptr()], s1, b1, None, 1, 0, True)
        o2 = conv(o1, wp[w2.data_ptr()], s2, b2, None, blk.stride, 1, True)
        if blk.downsample is not None:
            wd = _ohwi(blk.downsample[0].weight)
            sd, bd = blk.downsample[1].folded()
            idt = conv(x, wp[wd.data_ptr()], sd, bd, None, blk.stride, 0, False)
        else:
            idt = x
        x = conv(o2, wp[w3.data_ptr()], s3, b3, idt, 1, 0, True, out_=out if last else None)
    return out


def _backward_done(body) -> None:
    """the backbone node's backward has run (also replayed as a plan effect): the resident prefix buffer that forward read
    may be written again — see Backbone.features_nhwc"""
    body._bwd_pending = False
    if ops.L.RECORDER is not None:
        ops.L.RECORDER.effect(lambda: setattr(body, "_bwd_pending", False))


class _BackboneFn(Function):
    """frames [n,3,H,W] (NCHW, as handed over by the data pipeline) -> layer4 features NHWC [n,H/32,W/32,2048]."""

    @staticmethod
    def forward(ctx, frames, body: ResNet101Body, *weights):
        need_bwd = any(w.requires_grad for w in weights)
        s, b = body.bn1.folded()
        x = _stem(frames, body, s, b)
        x = ops.maxpool_raw(x)
        tape = []
        for li, blk in body.blocks():
            w1, w2, w3 = _ohwi(blk.conv1.weight), _ohwi(blk.conv2.weight), _ohwi(blk.conv3.weight)
            s1, b1 = blk.bn1.folded()
            s2, b2 = blk.bn2.folded()
            s3, b3 = blk.bn3.folded()
            o1 = ops.conv_fwd_raw(x, w1, s1, b1, None, 1, 0, True)
            o2 = ops.conv_fwd_raw(o1, w2, s2, b2, None, blk.stride, 1, True)
            wd = sd = None
            if blk.downsample is not None:
                wd = _ohwi(blk.downsample[0].weight)
                sd, bd = blk.downsample[1].folded()
                idt = ops.conv_fwd_raw(x, wd, sd, bd, None, blk.stride, 0, False)
            else:
                idt = x
            y = ops.conv_fwd_raw(o2, w3, s3, b3, idt, 1, 0, True)
            trainable = blk.conv1.weight.requires_grad
            if need_bwd and trainable:
                tape.append((blk, x, o1, o2, y, (w1, w2, w3, wd), (s1, s2, s3, sd)))
            x = y
        if tape and tape[-1][4] is x:       # the last block's output is this node's output: keep a detached alias (no cycle
            tape[-1] = tape[-1][:4] + (x.detach(),) + tape[-1][5:]      # through grad_fn: see _BackboneFnPl.forward)
        ctx.tape = tape
        ctx.body = body
        ctx.plist = weights
        return x

    @staticmethod
    def backward(ctx, dy):
        grads = {}
        tape = ctx.tape
        # transposed weights for every data-gradient GEMM of this pass: one launch (ops.WeightTransposer)
        ws = [w for rec in tape for w in rec[5] if w is not None]
        wts = ctx.body._wt_cache.refresh(ws)
        _wt = lambda w: wts.get(w.data_ptr())  # noqa: E731
        # weight gradients on a second stream behind the data-gradient chain (see _BackboneFnPl.backward)
        wg = ops.WgradStream(dy)

        def wgrad(param, g, xin, wshape, stride, pad):
            with wg:
                grads[id(param)] = ops.conv_wgrad_raw(g, xin, wshape, stride, pad)
            wg.keep(g, xin)

        blk, x, o1, o2, y, _, (s1, s2, s3, sd) = tape[-1]
        # top of the stack: dz = dy * [y > 0] (identity-path gradient), g3 = dz * scale3 (conv3 upstream)
        g3, dz = ops.act_bwd_raw(dy.contiguous(), y, s3, want_g=True, want_res=True, relu=True)
        for idx in range(len(tape) - 1, -1, -1):
            blk, x, o1, o2, y, (w1, w2, w3, wd), (s1, s2, s3, sd) = tape[idx]
            need_dx = idx > 0  # below the first trainable block everything is frozen (backbone.py:78-85)
            wgrad(blk.conv3.weight, g3, o2, w3.shape, 1, 0)
            # each dgrad epilogue applies the ReLU+BN backward of the layer below (no intermediate dO tensor)
            g2 = ops.conv_dgrad_raw(g3, w3, o2.shape, 1, 0, mask_y=o2, mask_scale=s2, wt=_wt(w3))
            wgrad(blk.conv2.weight, g2, o1, w2.shape, blk.stride, 1)
            g1 = ops.conv_dgrad_raw(g2, w2, o1.shape, blk.stride, 1, mask_y=o1, mask_scale=s1, wt=_wt(w2))
            wgrad(blk.conv1.weight, g1, x, w1.shape, 1, 0)
            gd = None
            if wd is not None:
                gd, _ = ops.act_bwd_raw(dz, None, sd, want_g=True, relu=False)  # dz * scale_downsample
                wgrad(blk.downsample[0].weight, gd, x, wd.shape, blk.stride, 0)
            if not need_dx:
                break
            # block boundary: x is the ReLU output of the block below; its dz / g3 come out of this epilogue
            s3_below = tape[idx - 1][6][2]
            if wd is not None:
                part = ops.conv_dgrad_raw(gd, wd, x.shape, blk.stride, 0, wt=_wt(wd))
                dz, g3 = ops.conv_dgrad_raw(g1, w1, x.shape, 1, 0, add=part, out=part, mask_y=x, scale2=s3_below, wt=_wt(w1))
            else:
                dz, g3 = ops.conv_dgrad_raw(g1, w1, x.shape, 1, 0, add=dz, mask_y=x, scale2=s3_below, wt=_wt(w1))
        wg.join(*grads.values())
        out = []
        for w in ctx.plist:  # same order as the *weights passed to forward
            g = grads.get(id(w))
            out.append(g.permute(0, 3, 1, 2) if g is not None else None)  # OHWI buffer seen as [O,I,KH,KW]
        if not getattr(ctx, "static", False):
            ctx.tape = None
        ops.dropout_backward_done()
        return (None, None, *out)


class _BackboneFnPl(Function):
    """The same network in mma mode "bf16x3p": every activation / gradient / weight between the max-pool and the layer4
    output is a pair of bf16 planes (ops.Planes) written by the producing kernel's epilogue and consumed by the
    LDS-DMA staged plane GEMMs (csrc/igemm_pl.h).  The layer4 output leaves as fp32 (consumer: input_proj)."""

    @staticmethod
    def forward(ctx, frames, body: ResNet101Body, pre_t, pre_mask, *weights):
        """pre_t / pre_mask: the frozen prefix's output for THESE frames (planes [NP, n, H/4, W/4, 256] + ReLU bit mask),
        computed under the previous step's grounding section (Backbone.stage_next); None = compute it here."""
        need_bwd = any(w.requires_grad for w in weights)
        blocks = list(body.blocks())
        n_pre = len(_prefix_blocks(body)) if pre_t is not None else 0
        if pre_t is None:
            s, b = body.bn1.folded()
            x = _stem(frames, body, s, b)
            x = ops.pl_maxpool_raw(x)
        else:
            x = ops.Planes(pre_t, pre_mask)
        # conv3 / downsample are followed by a FrozenBN and the block's ReLU: their upstream gradient is dz * scale.
        # The scale is folded into the transposed weight planes (data gradient) and into the weight-gradient epilogue,
        # so the backward pass works on dz alone and never writes a second, scaled copy of it.
        ws, ts = [], []
        for _, blk in blocks:
            ws += [_ohwi(blk.conv1.weight), _ohwi(blk.conv2.weight), _ohwi(blk.conv3.weight)]
            ts += [None, None, blk.bn3.folded()[0]]
            if blk.downsample is not None:
                ws.append(_ohwi(blk.downsample[0].weight))
                ts.append(blk.downsample[1].folded()[0])
        wp, wt = body._wpl_cache.refresh(ws, transposed=need_bwd, tscales=ts)
        tape = []
        yf = None
        # Two frame ranges, two streams (FORWARD_CHAINS): every conv of the forward pass is issued once per half of the
        # clip, the halves on their own streams, both writing into ONE whole-batch output (the backward pass is
        # unchanged).  A launch of the forward pass fills the 256 CUs unevenly (layer3, N = 256: 392 tiles = 1.53 rounds)
        # and alternates between an MFMA-bound K loop and an HBM-bound epilogue; with two independent chains the idle
        # CUs and the idle pipe of one launch are taken by the other chain's launch.
        n_all = x.shape[0]
        chains = _chain_cuts(n_all, x.t.is_cuda)
        sides = []
        if len(chains) > 1:
            sides = _chain_streams(x.t.device, len(chains))
            main = torch.cuda.current_stream(x.t.device)

        n_prefix_blocks = len(_prefix_blocks(body))
        pieces_pre = _prefix_ranges(chains)
        pieces_all = [(a, b, ci) for ci, (a, b) in enumerate(chains)]

        def conv(xin, w_, s_, b_, res, stride, pad, relu, planes_out=True, f32_out=False, want_mask=False, prefix=False):
            pieces = pieces_pre if prefix else pieces_all
            if not sides and len(pieces) == 1:
                return ops.pl_conv_fwd_raw(xin, w_, s_, b_, res, stride, pad, relu, planes_out=planes_out, f32_out=f32_out,
                                           want_mask=want_mask)
            n, H, W, _ = xin.shape
            Cout, KH = w_.shape[0], w_.shape[1]
            OH, OW = ops.conv_out_hw(H, W, KH, stride, pad)
            yp = ops.Planes.empty(xin.t, n, OH, OW, Cout) if planes_out else None
            yf_ = torch.empty(n, OH, OW, Cout, device=xin.device, dtype=torch.float32) if f32_out else None
            if want_mask and yp is not None:
                yp.mask = torch.empty(n * OH * OW, Cout // 8, device=xin.device, dtype=torch.uint8)
            for t_ in (yp.t if yp is not None else None, yp.mask if yp is not None else None, yf_):
                if t_ is not None:     # allocated on the main stream, written / read by the other chains as well
                    for sd_ in sides:
                        t_.record_stream(sd_)
                    if ops.L.RECORDER is not None:
                        ops.L.RECORDER.keep.append(t_)
            for a, b, ci in pieces:
                o = (yp.frames(a, b) if yp is not None else None, yf_[a:b] if yf_ is not None else None)
                args = (xin.frames(a, b), w_, s_, b_, res.frames(a, b) if res is not None else None, stride, pad, relu)
                if ci == 0 or not sides:
                    ops.pl_conv_fwd_raw(*args, out=o)
                else:
                    with torch.cuda.stream(sides[ci - 1]):
                        ops.pl_conv_fwd_raw(*args, out=o)
            return yp, yf_

        if sides:
            # Everything the chains READ must be queued on the main stream before they fork off it: on the first call the
            # FrozenBN folds (scale / bias per layer, cached afterwards) are launches of their own — folded inside the
            # block loop below they ran on the main stream while the side chain already read them (a first-step race
            # that a later warm step never shows; found by tests/test_dp_model.py once the stream probe moved the timing)
            for _, blk in blocks:
                blk.bn1.folded(), blk.bn2.folded(), blk.bn3.folded()
                if blk.downsample is not None:
                    blk.downsample[1].folded()
        for sd_ in sides:
            ops._wait_stream(sd_, main)     # the other chains start behind the max-pool / the weight planes / the folds
        for bi, (li, blk) in enumerate(blocks):
            if bi < n_pre:
                continue                   # the frozen prefix of these frames came in as pre_t
            last = bi == len(blocks) - 1
            w1, w2, w3 = _ohwi(blk.conv1.weight), _ohwi(blk.conv2.weight), _ohwi(blk.conv3.weight)
            s1, b1 = blk.bn1.folded()
            s2, b2 = blk.bn2.folded()
            s3, b3 = blk.bn3.folded()
            # ReLU bit masks for the backward pass: of o1 / o2 in trainable blocks, and of a block output whose
            # consumer (the next block) is trainable
            tr = need_bwd and blk.conv1.weight.requires_grad
            tr_next = need_bwd and not last and blocks[bi + 1][1].conv1.weight.requires_grad
            pre_ = bi < n_prefix_blocks       # a frozen prefix block computed in the step: the staged path's pieces
            o1, _ = conv(x, wp[w1.data_ptr()], s1, b1, None, 1, 0, True, want_mask=tr, prefix=pre_)
            o2, _ = conv(o1, wp[w2.data_ptr()], s2, b2, None, blk.stride, 1, True, want_mask=tr, prefix=pre_)
            wd = sd = None
            if blk.downsample is not None:
                wd = _ohwi(blk.downsample[0].weight)
                sd, bd = blk.downsample[1].folded()
                idt, _ = conv(x, wp[wd.data_ptr()], sd, bd, None, blk.stride, 0, False, prefix=pre_)
            else:
                idt = x
            y, yf = conv(o2, wp[w3.data_ptr()], s3, b3, idt, 1, 0, True, planes_out=not last, f32_out=last,
                         want_mask=tr_next, prefix=pre_)
            if need_bwd and blk.conv1.weight.requires_grad:
                # (the LAST block's output is this node's own output: the tape keeps a detached alias of it — the output
                #  object itself would close a reference cycle through its grad_fn that only backward() ever opened, and a
                #  forward that is never back-propagated (a validation pass without no_grad) leaked its whole tape: 5 GB at
                #  T = 16, found by the round-4 memory log of the GPU suite)
                tape.append((blk, x, o1, o2, yf.detach() if last else y, (w1, w2, w3, wd), (s1, s2, s3, sd)))
            x = y
        for sd_ in sides:
            ops._wait_stream(main, sd_)
        ctx.tape = tape
        ctx.wt = wt
        ctx.body = body
        ctx.plist = weights
        return yf

    @staticmethod
    def backward(ctx, dy):
        grads = {}
        tape, wt = ctx.tape, ctx.wt
        _wt = lambda w: wt[w.data_ptr()]  # noqa: E731
        # Weight gradients leave the critical path: they only feed the optimizer, so they run on a second stream
        # behind the data-gradient chain.  The plane GEMMs own a whole CU per workgroup and most launches fill the
        # chip unevenly (layer3: 196-224 workgroups on 256 CUs, a partial last round elsewhere): the weight-gradient
        # workgroups run on the CUs the data-gradient launch leaves idle.
        wg = ops.WgradStream(dy)
        sink = ops.GRAD_SINK
        delivered = set()

        # The data-gradient chain as two half-clip chains on two streams (BACKWARD_CHAINS), like the forward pass: every
        # data gradient is issued once per half into one whole-batch tensor; the weight gradients stay whole-batch
        # launches on the weight-gradient stream, ordered behind both halves.
        n_all = tape[-1][1].shape[0]
        fork = ops.fork_stream(dy) if (BACKWARD_CHAINS and n_all >= 8 and dy.is_cuda) else None
        chains = [(0, n_all // 2), (n_all // 2, n_all)] if (fork is not None and fork.active) else None

        def wgrad(param, g, xin, wshape, stride, pad, row_scale=None):
            if chains is not None:      # g's second half is written by the second chain
                ops._wait_stream(wg.side if wg.active else fork.main, fork.side)
            # data-parallel run: accumulate straight into the parameter's slot of its flat gradient bucket
            tgt = sink.grad_target(param) if sink is not None else None
            with wg:
                grads[id(param)] = ops.pl_conv_wgrad_raw(g, xin, wshape, stride, pad, row_scale,
                                                         out=tgt.permute(0, 2, 3, 1) if tgt is not None else None)
            wg.keep(g, xin)

        def dgrad(g, wt_, in_shape, k, stride, pad, add=None, out=None, mask_y=None, mask_scale=None):
            if chains is None:
                return ops.pl_conv_dgrad_raw(g, wt_, in_shape, k, stride, pad, add=add, out=out, mask_y=mask_y,
                                             mask_scale=mask_scale)
            dx = out
            if dx is None:
                dx = ops.Planes.empty(g.t, *in_shape)
                dx.t.record_stream(fork.side)
                if ops.L.RECORDER is not None:
                    ops.L.RECORDER.keep.append(dx.t)
            for ci, (a, b) in enumerate(chains):
                kw = dict(add=add.frames(a, b) if add is not None else None, out=dx.frames(a, b),
                          mask_y=mask_y.frames(a, b) if mask_y is not None else None, mask_scale=mask_scale)
                shp = (b - a,) + tuple(in_shape[1:])
                if ci == 0:
                    ops.pl_conv_dgrad_raw(g.frames(a, b), wt_, shp, k, stride, pad, **kw)
                else:
                    with torch.cuda.stream(fork.side):
                        ops.pl_conv_dgrad_raw(g.frames(a, b), wt_, shp, k, stride, pad, **kw)
            return dx

        def dgrad_cadd(g, wt_, in_shape, addc, add_stride, mask_y):
            if chains is None:
                return ops.pl_conv_dgrad_cadd_raw(g, wt_, in_shape, addc, add_stride, mask_y=mask_y)
            dx = ops.Planes.empty(g.t, *in_shape)
            dx.t.record_stream(fork.side)
            if ops.L.RECORDER is not None:
                ops.L.RECORDER.keep.append(dx.t)
            for ci, (a, b) in enumerate(chains):
                shp = (b - a,) + tuple(in_shape[1:])
                args = (g.frames(a, b), wt_, shp, addc.frames(a, b), add_stride)
                kw = dict(mask_y=mask_y.frames(a, b), out=dx.frames(a, b))
                if ci == 0:
                    ops.pl_conv_dgrad_cadd_raw(*args, **kw)
                else:
                    with torch.cuda.stream(fork.side):
                        ops.pl_conv_dgrad_cadd_raw(*args, **kw)
            return dx

        blk, x, o1, o2, y, _, (s1, s2, s3, sd) = tape[-1]
        # top of the stack (y is the fp32 layer4 output): dz = dy * [y > 0] as planes
        _, dz = ops.pl_act_bwd_raw(dy.contiguous(), y, None, want_g=False, want_res=True, relu=True)
        if chains is not None:
            dz.t.record_stream(fork.side)
            if ops.L.RECORDER is not None:
                ops.L.RECORDER.keep.append(dz.t)
            ops._wait_stream(fork.side, fork.main)
        for idx in range(len(tape) - 1, -1, -1):
            blk, x, o1, o2, y, (w1, w2, w3, wd), (s1, s2, s3, sd) = tape[idx]
            need_dx = idx > 0  # below the first trainable block everything is frozen (backbone.py:78-85)
            # conv3 (and the downsample conv) see dz directly: their FrozenBN scale sits in the transposed weight
            # planes and in the weight-gradient epilogue
            wgrad(blk.conv3.weight, dz, o2, w3.shape, 1, 0, s3)
            if wd is not None:
                # (round 6: the downsample conv's weight gradient needs dz and x only — queued here, not behind conv1's:
                #  in the LAST block of the pass, layer2.0, everything queued after the final data gradient is the step's
                #  exposed tail: tools/tail_probe.py, 0.64 ms of the main stream waiting at the last join)
                wgrad(blk.downsample[0].weight, dz, x, wd.shape, blk.stride, 0, sd)
            g2 = dgrad(dz, _wt(w3), o2.shape, 1, 1, 0, mask_y=o2, mask_scale=s2)
            wgrad(blk.conv2.weight, g2, o1, w2.shape, blk.stride, 1)
            g1 = dgrad(g2, _wt(w2), o1.shape, 3, blk.stride, 1, mask_y=o1, mask_scale=s1)
            wgrad(blk.conv1.weight, g1, x, w1.shape, 1, 0)
            if sink is not None:
                # data-parallel run: this block's weight gradients go to the gradient exchange now, from the weight-
                # gradient stream (behind the kernels that write them), not at the end of the whole backbone backward
                ws = [blk.conv3.weight, blk.conv2.weight, blk.conv1.weight]
                if wd is not None:
                    ws.append(blk.downsample[0].weight)
                gs = [grads[id(w_)].permute(0, 3, 1, 2) for w_ in ws]
                with wg:    # (a launch plan replays the hand-over at this point of the sequence, on this stream)
                    if ops.host_call(lambda ws=ws, gs=gs: sink.early(ws, gs)):
                        delivered.update(id(w_) for w_ in ws)
            if not need_dx:
                break
            # block boundary: x is the ReLU output of the block below; its dz comes out of this epilogue
            if wd is not None and blk.stride == 2 and COARSE_ADD and x.mask is not None:
                # round 5: the downsample conv's data gradient stays on ITS grid (a plain 1x1 GEMM on a quarter of the
                # pixels); conv1's data gradient adds it on the stride-2 lattice in its epilogue (round 4 scattered it
                # into a full-resolution tensor, 3/4 zeros, and read that back: 0.92 ms per launch at layer3.0 / layer4.0)
                n_, H_, W_, C_ = x.shape
                coarse = dgrad(dz, _wt(wd), (n_, dz.shape[1], dz.shape[2], C_), 1, 1, 0)
                dz = dgrad_cadd(g1, _wt(w1), x.shape, coarse, blk.stride, x)
            elif wd is not None:
                part = dgrad(dz, _wt(wd), x.shape, 1, blk.stride, 0)
                dz = dgrad(g1, _wt(w1), x.shape, 1, 1, 0, add=part, out=part, mask_y=x)
            else:
                dz = dgrad(g1, _wt(w1), x.shape, 1, 1, 0, add=dz, mask_y=x)
        if chains is not None:
            ops._wait_stream(fork.main, fork.side)
        wg.join(*grads.values())
        out = []
        for w in ctx.plist:
            g = grads.get(id(w)) if id(w) not in delivered else None
            out.append(g.permute(0, 3, 1, 2) if g is not None else None)
        if not getattr(ctx, "static", False):     # (a launch plan keeps the node's state: its tensors are static)
            ctx.tape = None
            ctx.wt = None
        ops.dropout_backward_done()
        _backward_done(ctx.body)
        return (None, None, None, None, *out)


class Backbone(nn.Module):
    """BackboneBase + Backbone (backbone.py:69-121): ``body`` holds the ResNet, layer2-4 are trainable."""

    def __init__(self, name: str = "resnet101", train_backbone: bool = True, return_interm_layers: bool = False,
                 dilation: bool = False):
        super().__init__()
        if name != "resnet101" or dilation or return_interm_layers:
            raise ValueError("stcat_amd implements the resnet101 / no-dilation / layer4-only configuration "
                             "selected by both reference experiment files")
        self.body = ResNet101Body()
        for n_, p in self.body.named_parameters():
            if not train_backbone or ("layer2" not in n_ and "layer3" not in n_ and "layer4" not in n_):
                p.requires_grad_(False)
        self.num_channels = 2048
        self._plist = None  # cached list(body.parameters()) (walking the module tree costs ~1 ms per step)
        self._staged = None      # (frames, version, ready event) of the clip the NEXT step will run (stage_next)
        self._prefix = None      # the frozen prefix computed for it: dict(frames, version, x, done, stream)
        self._pre_bufs = {}      # (shape, dtype, planes) -> two output plane sets, used alternately
        self._pre_plans = {}     # launch plans of the staged prefix, one per (geometry, buffer slot)
        self._pre_slot = 0
        self.prefix_stats = {"staged": 0, "taken": 0, "inline": 0}

    # ---- the next clip's frozen prefix under this step's grounding section (round 6) -----------------------------------
    # Stem + max-pool + layer1 are frozen (backbone.py:78-85): their output is a pure function of the frames and has no
    # backward.  At C3 they are ~4.6 ms at the head of a step's critical path, while the grounding section in the middle of
    # the step (encoder / decoders forward + backward, 14 ms) runs dependent chains of small launches that leave most of
    # the chip idle.  A training loop that knows its next clip (every data loader does: loader.DeviceFramePrefetcher has
    # it in HBM one step ahead) declares it with `stage_next`; once the CURRENT clip's forward has been queued, the next
    # clip's prefix is queued on a side stream, behind the current forward, into one of two resident plane buffers.  The
    # next step finds it, waits for its event and starts at layer2.  Nothing is cached across steps: every prefix is
    # computed once, from the frame buffer, for the one step that consumes it; a step whose frames were not staged (the
    # first step, an unmodified reference loop, evaluation) computes the prefix in place as before.
    _TRANSIENT = {"_staged": None, "_prefix": None, "_pre_bufs": None, "_plist": None, "_pre_plans": None}

    def __getstate__(self):
        """copy.deepcopy / pickling (the EMA copy of scripts/train_net.py:62-64, torch.save(model)): the staging state — the
        declared frames, the computed prefix with its HIP event, the two resident plane buffers — belongs to the running
        loop, not to the module (an Event cannot be pickled, 2.4 GB of buffers should not be cloned into an EMA model)"""
        st = dict(self.__dict__)
        for k in self._TRANSIENT:
            st[k] = {} if k in ("_pre_bufs", "_pre_plans") else None
        st["prefix_stats"] = {"staged": 0, "taken": 0, "inline": 0}
        return st

    def stage_next(self, frames: torch.Tensor, ready=None) -> None:
        """declare the frames the NEXT call of the model will see (device tensor; fp32 [n,3,H,W] or uint8 [n,H,W,3]);
        `ready`: event after which they are valid (the copy stream's, loader.DeviceFramePrefetcher)"""
        if frames is None or not PREFIX_PIPELINE:
            self._staged = None
            return
        self._staged = (frames, frames._version, ready)
        self.prefix_stats["staged"] += 1

    def _fill(self) -> None:
        """called once the current clip's forward has been queued — from the query decoder's entry (ops.run_deferred) or, with
        STCAT_PREFIX_AT=backbone, right behind the backbone node: queue the staged clip's prefix on its lane"""
        st, self._staged = self._staged, None
        if st is None or not ops.L.plane_count() or not _prefix_blocks(self.body):
            return
        frames, ver, ready = st
        if frames._version != ver:
            return                           # rewritten since it was declared: the consuming step computes in place
        dev = frames.device
        cuda = dev.type == "cuda"
        if cuda and torch.cuda.is_current_stream_capturing():
            return
        blks = _prefix_blocks(self.body)
        n = frames.shape[0]
        H, W = (frames.shape[1], frames.shape[2]) if frames.dtype == torch.uint8 else (frames.shape[2], frames.shape[3])
        OH, OW = ops.conv_out_hw(*ops.conv_out_hw(H, W, 7, 2, 3), 3, 2, 1)
        C = blks[-1].conv3.weight.shape[0]
        want_mask = len(blks) < sum(BLOCKS) and list(self.body.blocks())[len(blks)][1].conv1.weight.requires_grad
        key = (n, OH, OW, C, ops.L.plane_count(), str(dev), want_mask)
        bufs = self._pre_bufs.get(key)
        main = torch.cuda.current_stream(dev) if cuda else None
        if bufs is None:
            if cuda and self._pre_bufs and str(dev) in _PREFIX_LANES:
                main.wait_stream(_PREFIX_LANES[str(dev)])     # (a prefix of the old geometry may still be running on its lane)
            self._pre_bufs.clear()           # (one clip geometry at a time: 1.2 GB per buffer at C3)
            self._pre_plans.clear()
            bufs = []
            for _ in range(2):
                pl = ops.Planes.empty(frames, n, OH, OW, C)
                if want_mask:
                    pl.mask = torch.empty(n * OH * OW, C // 8, device=dev, dtype=torch.uint8)
                bufs.append(pl)
            self._pre_bufs[key] = bufs
        self._pre_slot ^= 1
        out = bufs[self._pre_slot]
        # The buffer's previous user is the step BEFORE the current one (two buffers, alternating): its backward pass —
        # layer2.0's weight gradients read the prefix output on the weight-gradient stream, joined into the main stream at
        # the end of the backward node — precedes this point on the main stream.
        side = None
        if cuda and ops.FORK_ENABLED and ops.L._backend == "hip":
            side = _prefix_lane(dev)
            if side.cuda_stream == main.cuda_stream:
                side = None
        with torch.no_grad():
            if side is None:
                self._run_prefix(frames, out, key)
                done = None
            else:
                side.wait_stream(main)               # behind the current clip's forward (and the weight-plane refresh)
                if ready is not None:
                    side.wait_event(ready)
                with torch.cuda.stream(side):        # (its temporaries belong to the side stream's pool)
                    self._run_prefix(frames, out, key)
                    done = torch.cuda.Event()
                    done.record(side)
                frames.record_stream(side)
        self._prefix = {"frames": frames, "version": ver, "x": out, "done": done, "want_mask": want_mask,
                        "state": self._prefix_state()}

    def _run_prefix(self, frames, out, key) -> None:
        """the staged prefix's launches on the current stream: eagerly the first time a (geometry, buffer slot) is seen,
        recorded into a launch plan the second time, replayed by ONE C call afterwards (its 22 — or, in pieces of
        _prefix_range() frames, ~170 — launches otherwise cost the host thread 0.5 — 4 ms per step, which shows as soon as
        the host is busy: with a live process group the step is nearly host-bound).  The plan's only external tensor is
        the frame buffer; its output planes are the resident buffer of this slot, its temporaries live in its own pool."""
        if not plans.ENABLED or ops.L.RECORDER is not None or frames.device.type != "cuda" or PREFIX_EAGER:
            _prefix_forward(frames, self.body, out)
            return
        pk = (key, self._pre_slot, tuple(frames.shape), frames.dtype, frames.data_ptr() % 16, self._prefix_state())
        if not any(c is self._pre_plans for c in plans._CACHES):
            plans._CACHES.append(self._pre_plans)        # (plans.clear() drops these with every other plan)
        ent = self._pre_plans.get(pk)
        if ent is None:
            if len(self._pre_plans) >= 8:
                self._pre_plans.clear()
            ent = self._pre_plans[pk] = {"calls": 0, "plan": None, "pool": None, "refused": False}
        ent["calls"] += 1
        if ent["calls"] == 1 or ent["refused"]:
            _prefix_forward(frames, self.body, out)
        elif ent["plan"] is None:
            ent["pool"] = torch.cuda.MemPool()
            _, plan = plans._record(frames.device, [frames], ent["pool"], lambda: _prefix_forward(frames, self.body, out))
            if plan is None:
                ent["refused"] = True
            ent["plan"] = plan
        else:
            ent["plan"].run([frames])

    def _prefix_state(self):
        """what a computed prefix depends on besides the frames: the arithmetic mode (plane count AND element type), the
        static tables (FrozenBN folds / weight planes: plans.STATIC_EPOCH moves with load_state_dict / .to()) and how many
        blocks are frozen"""
        return (ops.L.get_mma_mode(), ops.L.plane_count(), plans.STATIC_EPOCH, len(_prefix_blocks(self.body)), _prefix_range())

    def _take(self, frames: torch.Tensor):
        """the prefix computed for exactly these frames, or None; the current stream is ordered behind it"""
        pre, self._prefix = self._prefix, None
        if pre is None:
            return None
        if (pre["frames"] is not frames and not (pre["frames"].data_ptr() == frames.data_ptr()
                                                   and pre["frames"].shape == frames.shape
                                                   and pre["frames"].dtype == frames.dtype)) \
                or frames._version != pre["version"] or pre["state"] != self._prefix_state():
            return None
        if pre["done"] is not None:
            torch.cuda.current_stream(frames.device).wait_event(pre["done"])
        return pre["x"]

    def features_nhwc(self, frames: torch.Tensor) -> torch.Tensor:
        ops.drop_deferred()      # (a fill that the previous pass never reached — a model without our decoder: dropped)
        if self.training and torch.is_grad_enabled():
            # first module of the hot path to run in a step: open the step's dropout counter range (drop-in mode has
            # no other place to do it — the reference's train loop is unmodified)
            ops.dropout_auto_begin_step(frames.device)
            ops.dropout_forward_started()
        weights = self._plist  # same order as body.parameters(): the backward returns one gradient per entry
        if weights is None or len(weights) == 0:
            weights = self._plist = [p for p in self.body.parameters()]
        if ops.L.plane_count():
            pre = self._take(frames)
            need_mask = torch.is_grad_enabled() and any(w.requires_grad for w in weights)
            if pre is not None and need_mask and pre.mask is None:
                pre = None
            # Two resident buffers alternate: this forward reads one (its backward too: layer2.0's weight gradients), the
            # prefix staged during this forward fills the other.  That is only safe when the PREVIOUS forward's backward
            # has run — with two forwards before a backward (gradient accumulation over two clips) the staged prefix
            # would overwrite the buffer the earlier forward's backward still needs: no staging in such a pass (the next
            # step computes its prefix in place).  A forward that is never back-propagated costs one skipped staging.
            if need_mask:
                if getattr(self.body, "_bwd_pending", False) and self._staged is not None:
                    self._staged = None
                    self.prefix_stats["skipped_busy"] = self.prefix_stats.get("skipped_busy", 0) + 1
                self.body._bwd_pending = True
            self.prefix_stats["taken" if pre is not None else "inline"] += 1
            feat = plans.apply(_BackboneFnPl, frames, self.body, pre.t if pre is not None else None,
                               pre.mask if pre is not None else None, *weights)
            # the NEXT clip's frozen prefix, if one was declared (stage_next): behind this forward — at once, or handed to
            # the query decoder's entry (ops.run_deferred in QueryDecoder.run), where the chip has room for it
            if PREFIX_AT == "decoder" and self._staged is not None:
                ops.defer(self._fill)
            else:
                self._fill()
            return feat
        self._staged = None
        return plans.apply(_BackboneFn, frames, self.body, *weights)

    def forward(self, tensor_list: NestedTensor):
        feat = self.features_nhwc(tensor_list.tensors)
        m = tensor_list.mask
        assert m is not None
        mask = F.interpolate(m[None].float(), size=feat.shape[1:3]).to(torch.bool)[0]  # backbone.py:100
        return {"0": NestedTensor(feat.permute(0, 3, 1, 2), mask, tensor_list.durations)}


class PositionEmbeddingSine(nn.Module):
    """vision_model/position_encoding.py:51-94 with num_pos_feats=128, normalize=True (built at :138)."""

    def __init__(self, num_pos_feats: int = 128, temperature: int = 10000, normalize: bool = True, scale=None):
        super().__init__()
        if num_pos_feats != 128 or temperature != 10000 or not normalize:
            raise ValueError("only PositionEmbeddingSine(128, normalize=True) is implemented")

    def tokens(self, mask: torch.Tensor) -> torch.Tensor:
        return ops.pos_sine_2d(mask)  # [n, h*w, 256]

    def forward(self, tensor_list: NestedTensor) -> torch.Tensor:
        n, h, w = tensor_list.mask.shape
        return self.tokens(tensor_list.mask).view(n, h, w, 256).permute(0, 3, 1, 2)


class Joiner(plans.InvalidatesPlans, nn.Sequential):
    """backbone.py:147-159."""

    def __init__(self, backbone: Backbone, position_embedding: PositionEmbeddingSine):
        super().__init__(backbone, position_embedding)
        self.num_channels = backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        out = self[0](tensor_list)["0"]
        return out, self[1](out).to(out.tensors.dtype)

    def forward_tokens(self, frames: torch.Tensor, mask: torch.Tensor):
        """internal fast path: NHWC features, layer4-resolution mask, token-major positions."""
        feat = self[0].features_nhwc(frames)
        m = F.interpolate(mask[None].float(), size=feat.shape[1:3]).to(torch.bool)[0]
        return feat, m, self[1].tokens(m)


def build_vis_encoder(cfg=None) -> Joiner:
    """Counterpart of models/vision_model/__init__.py:5-24."""
    train_backbone, name, dilation, pos_enc = True, "resnet101", False, "sine"
    if cfg is not None:
        train_backbone = cfg.SOLVER.VIS_BACKBONE_LR > 0
        name = cfg.MODEL.VISION_BACKBONE.NAME
        dilation = cfg.MODEL.VISION_BACKBONE.DILATION
        pos_enc = cfg.MODEL.VISION_BACKBONE.POS_ENC
        if cfg.MODEL.STCAT.HIDDEN != 256:
            raise ValueError("HIDDEN must be 256")
    if pos_enc != "sine":
        raise ValueError(f"not supported {pos_enc}")  # position_encoding.py:144
    return Joiner(Backbone(name, train_backbone, False, dilation), PositionEmbeddingSine(128, normalize=True))
