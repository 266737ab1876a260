"""Autograd operators of the STCAT hot path, each a thin host wrapper around the
HIP C ABI (include/stcat_hip.h).  PyTorch supplies device memory, streams and the
autograd tape; every arithmetic kernel is ours.  Token tensors are row-major
[rows, features]; images are NHWC.  There is no fallback: if the library is
missing or a tensor is not on the GPU the call raises.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import _lib as L

_f32 = torch.float32


def _chk(*ts):
    for t in ts:
        if t is not None:
            L.check_tensor(t)
            if t.dtype != _f32 and t.dtype not in (torch.uint8, torch.bool, torch.int32):
                raise L.StcatHipError(f"unsupported dtype {t.dtype}")


def _empty(like: torch.Tensor, *shape) -> torch.Tensor:
    return torch.empty(shape, device=like.device, dtype=_f32)


class ZeroArena:
    """One flat, pre-zeroed fp32 buffer that serves every zero-initialised temporary of a step (weight-gradient
    accumulators of the split-K wgrad kernels, bias / LayerNorm gradient sums): a single memset per step replaces
    ~800 tiny fill launches.  `reset()` must be called once per step before the first backward op."""

    def __init__(self, device, numel: int):
        self.buf = torch.zeros(numel, device=device, dtype=_f32)
        self.off = 0
        self.high = 0

    def reset(self):
        if self.off:
            self.buf[: self.off].zero_()
        self.high = max(self.high, self.off)
        self.off = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        start = (self.off + 63) & ~63  # 256-byte alignment
        if start + n > self.buf.numel():
            return None
        self.off = start + n
        return self.buf[start:start + n].view(*shape)


_ARENA = {}


def enable_zero_arena(device, numel: int) -> ZeroArena:
    a = ZeroArena(device, numel)
    _ARENA[str(device)] = a
    return a


def disable_zero_arena():
    _ARENA.clear()


def _zeros(like: torch.Tensor, *shape) -> torch.Tensor:
    if L.RECORDER is not None:          # a launch plan is being recorded: its own chunks, cleared at the top of a replay
        return L.RECORDER.zeros(shape)
    a = _ARENA.get(str(like.device))
    if a is not None:
        t = a.take(shape)
        if t is not None:
            return t
    return torch.zeros(shape, device=like.device, dtype=_f32)


def _zero_take(like: torch.Tensor, shape) -> Optional[torch.Tensor]:
    """a zeroed buffer that costs no fill launch (the recording plan's chunks, else the step's arena), or None"""
    if L.RECORDER is not None:
        return L.RECORDER.zeros(shape)
    a = _ARENA.get(str(like.device))
    return a.take(shape) if a is not None else None


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else (t if t.is_contiguous() else t.contiguous())


_DIMT = {}


def dim_t(device) -> torch.Tensor:
    """10000^(2*floor(i/2)/128), i < 128, computed in fp32 on the host exactly as the
    reference does (net_utils.py:34-35, vision_model/position_encoding.py:82-84)."""
    key = str(device)
    if key not in _DIMT:
        i = torch.arange(128, dtype=torch.float32)
        _DIMT[key] = (10000 ** (2 * torch.div(i, 2, rounding_mode="floor") / 128)).to(device)
    return _DIMT[key]


# ------------------------------------------------------------------------------------
# second HIP stream for independent launch chains
# ------------------------------------------------------------------------------------
_SIDE_STREAMS = {}
FORK_ENABLED = not os.environ.get("STCAT_NO_FORK")  # two-stream decoders (QueryDecoder.run)
GRAD_SINK = None  # dist.GradBucketReducer when gradients are exchanged: .early(params, grads) takes them mid-backward


STREAM_PROBE = not os.environ.get("STCAT_NO_STREAM_PROBE")
_PICKED = {}          # device -> [side stream, weight-gradient stream, ...]
PICK_REPORT = {}      # device -> what the probe saw (bench.py prints it)


def _pick_streams(dev):
    """Side streams that REALLY run beside the current stream.  HIP maps all streams of a process onto GPU_MAX_HW_QUEUES
    (default 4) hardware queues per priority; two streams that share a queue serialise, and which ones share depends on
    how many streams the framework / RCCL created before us (measured: with a live RCCL process group the second forward
    chain landed on the main stream's queue — the backbone forward lost its overlap, +3.5 ms per C3 step; raising
    GPU_MAX_HW_QUEUES instead made the step 25 % slower; profiles/r03_hw_queues.log).  So the choice is MEASURED once per
    device: candidates from torch's stream pool are timed against the current stream and against each other with a
    500 us one-workgroup spin kernel (stcat_spin), host-timed — concurrent pairs take ~0.5 ms, queue-sharing pairs ~1 ms —
    and the first (up to) three pairwise-concurrent ones are kept: [side, weight-gradient, spare].  ~40 ms once, on the
    first eager call (never inside a plan recording or a graph capture)."""
    got = _PICKED.get(dev)
    if got is not None:
        return got
    cands = [torch.cuda.Stream(device=dev) for _ in range(12)]
    picked, report = cands[:2], {"probed": False}
    main = torch.cuda.current_stream(dev)
    if (STREAM_PROBE and L._backend == "hip" and dev.type == "cuda" and L.RECORDER is None
            and not torch.cuda.is_current_stream_capturing()):
        lib = L.load()
        us = 500

        import time

        def run_ms(streams):
            # host-timed, no cross-stream events: device idle -> launch on every stream -> device idle again (event waits
            # between queues add ~0.2 ms of their own inside a busy process and blur the 1x / 2x answer)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for st in streams:
                if lib.stcat_spin(us, st.cuda_stream) != 0:
                    raise L.StcatHipError("stcat_spin failed")
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) * 1e3

        def pair_ms(a, b):
            return min(run_ms((a, b)), run_ms((a, b)))

        def one_ms(a):
            return run_ms((a,))

        pair_ms(main, cands[0])                       # warm-up (first launch on a fresh stream)
        one_ms(main)
        single = min(one_ms(main), one_ms(main))      # the spin's real length (the counter's rate is not assumed)
        conc = lambda a, b: pair_ms(a, b) < 1.5 * single   # noqa: E731
        with_main = [pair_ms(main, c) for c in cands]
        beside_main = [c for c, ms in zip(cands, with_main) if ms < 1.5 * single]
        chosen = []                                   # greedy: pairwise-concurrent streams beside the current one
        for c in beside_main:
            if len(chosen) == 3:
                break
            if all(conc(c, o) for o in chosen):
                chosen.append(c)
        if chosen:
            picked = list(chosen)
        report = {"probed": True, "candidates": len(cands), "concurrent_with_main": len(beside_main),
                  "picked": [cands.index(c) for c in chosen], "spin_ms": round(single, 3),
                  "pair_ms_with_main": [round(x, 3) for x in with_main]}
    while len(picked) < 2:                            # (fewer than two queues beside the current one: roles share a stream)
        picked.append(picked[-1])
    _PICKED[dev] = picked
    PICK_REPORT[str(dev)] = report
    return picked


def side_stream(dev, index: int = 0):
    """the package's side streams, by index: 0 serves the forward chains of the backbone AND the forked time decoder
    (never busy together); 1 is the weight-gradient stream (WgradStream).  Main + side + weight-gradient (+ RCCL's) fit
    the four hardware queues HIP gives a process by default; _pick_streams makes sure they really sit on different ones."""
    got = _pick_streams(dev)
    if index < len(got):
        return got[index]
    have = _SIDE_STREAMS.setdefault(dev, [])          # beyond the measured ones: may share a hardware queue
    while len(have) <= index - len(got):
        have.append(torch.cuda.Stream(device=dev))
    return have[index - len(got)]


def _wait_stream(waiter, signal) -> None:
    """waiter.wait_stream(signal); mirrored into the launch plan that is being recorded, if any"""
    waiter.wait_stream(signal)
    if L.RECORDER is not None:
        L.RECORDER.wait(waiter, signal)


# ------------------------------------------------------------------------------------
# work queued by one module of the path for a later point of the same forward pass
# ------------------------------------------------------------------------------------
_DEFERRED = []


def defer(fn) -> None:
    """`fn()` runs at the next `run_deferred()` — the visual encoder hands the next clip's frozen prefix (Backbone._fill) to
    the query decoder's entry this way: launched there, it runs under the decoders' latency-bound chains instead of
    competing with the encoder's full-chip GEMMs (measured: issued right behind the backbone it doubled the encoder's
    forward, 3.3 -> 6.5 ms; profiles/r06_prefix_pipeline.log)"""
    _DEFERRED.append(fn)


def run_deferred() -> None:
    while _DEFERRED:
        _DEFERRED.pop(0)()


def drop_deferred() -> None:
    _DEFERRED.clear()


def host_call(fn):
    """run a host-side action that belongs at this point of a node's launch sequence (a launch plan replays it here)"""
    if L.RECORDER is not None:
        return L.RECORDER.host_call(fn)
    return fn()


class fork_stream:
    """`with fork_stream(x): ...` runs the body on a side stream ordered after everything queued so far on the
    current stream; `join(*outputs)` makes the current stream wait for it and tells the caching allocator that the
    outputs are used on the current stream from now on.  A no-op on the emulator backend / when disabled."""

    def __init__(self, like: torch.Tensor):
        # Measured at C3 (bench.py, round 2): single GPU 65.2 -> 62.8 ms with the fork; with the gradient exchange active
        # (STCAT_FORCE_COMM=1: RCCL stream + bucket copies) 70.3 -> 66.3 ms — N = 1 and N > 1 run the same schedule.
        self.active = FORK_ENABLED and L._backend == "hip" and like.is_cuda
        if self.active:
            dev = like.device
            self.main = torch.cuda.current_stream(dev)
            self.side = side_stream(dev, 0)
            self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        if self.active:
            _wait_stream(self.side, self.main)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *outputs):
        if self.active:
            _wait_stream(self.main, self.side)
            for t in outputs:
                if torch.is_tensor(t):
                    t.record_stream(self.main)


# ------------------------------------------------------------------------------------
# raw (no autograd) wrappers
# ------------------------------------------------------------------------------------
def ew(op: int, a, b=None, c=None, bmod: int = 0, alpha: float = 1.0, beta: float = 1.0, out=None):
    a = _c(a)
    b = _c(b)
    c = _c(c)
    _chk(a, b, c)
    out = torch.empty_like(a) if out is None else out
    L.call("stcat_ew", op, a.data_ptr(), L._ptr(b), L._ptr(c), out.data_ptr(), a.numel(), bmod, alpha, beta,
           L.stream_of(a))
    return out


def ew2d(op: int, a, b=None, out=None, alpha: float = 1.0, beta: float = 1.0):
    """two-operand op / copy on [..., C] operands whose rows may be strided (column blocks of wider matrices)"""
    C = a.shape[-1]

    def rows(t):
        if t.dim() == 2:
            t2 = t
        elif t.is_contiguous():
            t2 = t.view(-1, C)
        else:
            t2 = t.reshape(-1, C) if t.stride(-1) == 1 and all(
                t.stride(i) == t.stride(i + 1) * t.shape[i + 1] for i in range(t.dim() - 2)) else _c(t).view(-1, C)
        assert t2.stride(1) == 1
        return t2
    a2 = rows(a)
    b2 = rows(b) if b is not None else None
    ret = out if out is not None else torch.empty(a.shape, device=a.device, dtype=a.dtype)
    o2 = rows(ret)
    assert o2.data_ptr() == ret.data_ptr()
    _chk(a2, b2)
    R = o2.shape[0]            # a one-row operand is broadcast over the output rows (leading dimension 0)
    lda = a2.stride(0) if a2.shape[0] == R else 0
    ldb = 0 if b2 is None or b2.shape[0] != R else b2.stride(0)
    assert a2.shape[0] in (1, R) and (b2 is None or b2.shape[0] in (1, R))
    L.call("stcat_ew2d", op, a2.data_ptr(), lda, L._ptr(b2), ldb, o2.data_ptr(), o2.stride(0), R, C, alpha, beta,
           L.stream_of(a2))
    return ret


def colsum(a2d: torch.Tensor, b2d: Optional[torch.Tensor] = None) -> torch.Tensor:
    M, N = a2d.shape
    out = _zeros(a2d, N)
    L.call("stcat_colsum", a2d.data_ptr(), L._ptr(b2d), out.data_ptr(), M, N, L.stream_of(a2d))
    return out


def linear_fwd_raw(x2d, w, bias, res2d=None, relu=False, out=None, ldy=None, c_group=0, c_group_stride=0):
    M, K = x2d.shape
    N = w.shape[0]
    _chk(x2d, w, bias, res2d)
    if out is None:
        # the decoders' skinny launches ([T,256] states): output taken from the step's zeroed arena, reduction split
        # over grid.z with an atomic epilogue — 8 serial K-tiles on 4 workgroups are pure latency otherwise
        if (M <= 128 and K >= 128 and K % 64 == 0 and N % 64 == 0 and not relu and c_group == 0 and ldy is None
                and L.get_mma_mode() != "f32" and x2d.is_contiguous()):
            out = _zero_take(x2d, (M, N))
            if out is not None:
                L.call("stcat_linear_fwd_acc", x2d.data_ptr(), w.data_ptr(), L._ptr(bias), L._ptr(res2d), out.data_ptr(), M, N, K,
                       K, N, (res2d.stride(0) if res2d is not None else 0), L.stream_of(x2d))
                return out
        out = _empty(x2d, M, N)
    ldy = N if ldy is None else ldy
    if N % 64 == 0:
        L.call("stcat_linear_fwd", x2d.data_ptr(), w.data_ptr(), L._ptr(bias), L._ptr(res2d), out.data_ptr(), M, N, K,
               x2d.stride(0), ldy, (res2d.stride(0) if res2d is not None else 0), int(relu), c_group, c_group_stride,
               L.stream_of(x2d))
    else:
        assert res2d is None and not relu and c_group == 0 and x2d.is_contiguous()
        L.call("stcat_small_linear_fwd", x2d.data_ptr(), w.data_ptr(), L._ptr(bias), out.data_ptr(), M, N, K,
               L.stream_of(x2d))
    return out


def _pad8(xs):
    xs = [L._ptr(x) for x in xs]
    return xs + [None] * (8 - len(xs))


def linear_multi_ok(M: int, N: int, K: int, like: torch.Tensor) -> bool:
    """the grouped skinny-Linear launches apply: a split-bf16 mode, M <= 128, 128-multiples (stcat_linear_wgrad_multi's
    tile, dense leading dimensions), and a source of zeroed outputs"""
    return (MULTI_LINEAR and L.get_mma_mode() != "f32" and M <= 128 and N % 128 == 0 and K % 128 == 0
            and (L.RECORDER is not None or _ARENA.get(str(like.device)) is not None))


def linear_fwd_multi(xs, ws, bs, ys, M, N, K):
    """ys[j] += xs[j] ws[j]^T + bs[j] for up to 8 problems of one shape in ONE launch (ys zeroed by the caller, may coincide)"""
    n = len(xs)
    assert 1 <= n <= 8 and len(ws) == n and len(bs) == n and len(ys) == n
    L.call("stcat_linear_fwd_multi", n, *_pad8(xs), *_pad8(ws), *_pad8(bs), *_pad8(ys), M, N, K, L.stream_of(xs[0]))


def linear_dgrad_multi(gs, ws, adds, dxs, M, N, K):
    """dxs[j] += gs[j] ws[j] (+ adds[j]); ws[j] is [N, K]"""
    n = len(gs)
    assert 1 <= n <= 8
    L.call("stcat_linear_dgrad_multi", n, *_pad8(gs), *_pad8(ws), *_pad8(adds), *_pad8(dxs), M, N, K, L.stream_of(gs[0]))


def linear_wgrad_multi(gs, xs, dws, dbs, M, N, K):
    """dws[j] += gs[j]^T xs[j], dbs[j] += column sums of gs[j]"""
    n = len(gs)
    assert 1 <= n <= 8
    L.call("stcat_linear_wgrad_multi", n, *_pad8(gs), *_pad8(xs), *_pad8(dws), *_pad8(dbs), M, N, K, L.stream_of(gs[0]))


MULTI_LINEAR = not os.environ.get("STCAT_NO_MULTI_LINEAR")


def act_bwd_raw(dy, y, scale, want_g=True, want_res=False, relu=True):
    dy = _c(dy)
    n = dy.numel()
    C = dy.shape[-1]
    G = torch.empty_like(dy) if want_g else None
    R = torch.empty_like(dy) if want_res else None
    L.call("stcat_act_bwd", dy.data_ptr(), L._ptr(y), L._ptr(scale), L._ptr(G), L._ptr(R), n, C, int(relu),
           L.stream_of(dy))
    return G, R


# ------------------------------------------------------------------------------------
# VideoSTGLoss: all decoder layers, all five terms, one launch each way (csrc/loss.h)
# ------------------------------------------------------------------------------------
LOSS_ROWS = ("loss_bbox", "loss_giou", "loss_sted", "loss_guided_attn", "loss_actioness")


class StgLossFn(Function):
    """(boxes [nl,rows,4], sted [nl,b,T,2], weights [nl,b,T,T], act [nl,b,T]|None, plan, wmat [5,nl]|None) ->
    (vec [5,nl] un-weighted losses per layer, total = sum wmat*vec | None) — models/criterion.py:11-208."""

    @staticmethod
    def _args(boxes, sted, w, act, plan):
        nl, rows_total = boxes.shape[0], boxes.shape[1]
        nb = plan.num_boxes(boxes.device)
        nb_dev = nb if torch.is_tensor(nb) else None
        return (boxes.data_ptr(), plan.rows.data_ptr(), plan.tgt_boxes.data_ptr(), sted.data_ptr(), plan.dist.data_ptr(),
                plan.time_mask_u8.data_ptr(), w.data_ptr(), plan.pos_or_pad_u8.data_ptr(), plan.nb_neg.data_ptr(),
                L._ptr(act), plan.actioness.data_ptr(), plan.act_weight.data_ptr(), L._ptr(nb_dev),
                0.0 if nb_dev is not None else float(nb), nl, rows_total, plan.rows.numel(), plan.b, plan.T)

    @staticmethod
    def forward(ctx, boxes, sted, w, act, plan, wmat):
        boxes, sted, w, act = _c(boxes), _c(sted), _c(w), _c(act)
        _chk(boxes, sted, w, act, wmat)
        nl = boxes.shape[0]
        assert sted.shape == (nl, plan.b, plan.T, 2) and w.shape == (nl, plan.b, plan.T, plan.T)
        # the kernels index boxes / targets / logits with the plan's rows: a mismatch must raise here, as the reference's
        # F.l1_loss does, not read out of bounds
        assert boxes.dim() == 3 and boxes.shape[2] == 4 and boxes.shape[1] == plan.b * plan.T, (boxes.shape, plan.b, plan.T)
        assert tuple(plan.tgt_boxes.shape) == (plan.rows.numel(), 4), (plan.tgt_boxes.shape, plan.rows.numel())
        assert act is None or tuple(act.shape) == (nl, plan.b, plan.T), act.shape
        vec = _empty(boxes, 5, nl)
        total = _zeros(boxes, 1) if wmat is not None else None
        L.call("stcat_stg_loss_fwd", *StgLossFn._args(boxes, sted, w, act, plan), L._ptr(wmat), vec.data_ptr(), L._ptr(total),
               L.stream_of(boxes))
        ctx.save_for_backward(boxes, sted, w, act, wmat)
        ctx.plan = plan
        ctx.set_materialize_grads(False)
        if total is None:
            ctx.mark_non_differentiable()
            return vec, None
        return vec, total.view(())

    @staticmethod
    def backward(ctx, gvec, gtotal):
        boxes, sted, w, act, wmat = ctx.saved_tensors
        if gvec is None and gtotal is None:
            return None, None, None, None, None, None
        LINEAR_WT.refresh_all(boxes)      # the first backward node of a step: every Linear W^T of the step in one launch
        gvec, gtotal = _c(gvec), _c(gtotal)
        d_boxes, d_sted, d_w = torch.empty_like(boxes), torch.empty_like(sted), torch.empty_like(w)
        d_act = torch.empty_like(act) if act is not None else None
        L.call("stcat_stg_loss_bwd", *StgLossFn._args(boxes, sted, w, act, ctx.plan), L._ptr(wmat), L._ptr(gvec),
               L._ptr(gtotal), d_boxes.data_ptr(), d_sted.data_ptr(), d_w.data_ptr(), L._ptr(d_act), L.stream_of(boxes))
        return d_boxes, d_sted, d_w, d_act, None, None


# ------------------------------------------------------------------------------------
# Linear (+bias, +residual, +ReLU)
# ------------------------------------------------------------------------------------
class LinearFn(Function):
    """y = relu?(x W^T + b + res)  — torch.nn.Linear call sites of the grounding model
    (modal_encoder.py:214-216, query_decoder.py:262-284, net_utils.py:13-15)."""

    @staticmethod
    def forward(ctx, x, w, b, res, relu):
        shp = x.shape
        x2 = _c(x.reshape(-1, shp[-1]))
        w = _c(w)
        r2 = _c(res.reshape(-1, w.shape[0])) if res is not None else None
        y = linear_fwd_raw(x2, w, _c(b), r2, relu)
        ctx.relu = relu
        ctx.has_res = res is not None
        ctx.has_b = b is not None
        ctx.save_for_backward(x2, w, y if relu else None)
        ctx.xshape = shp
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        N, K = w.shape
        g = _c(dy.reshape(-1, N))
        if ctx.relu:
            g, _ = act_bwd_raw(g, y, None, want_g=True, relu=True)
        M = g.shape[0]
        st = L.stream_of(g)
        dx = dw = db = None
        if N % 64 == 0:
            if ctx.needs_input_grad[0]:
                dx = _empty(g, M, K)
                wt = LINEAR_WT.get(w) if M > 256 else None
                L.call("stcat_linear_dgrad", g.data_ptr(), w.data_ptr(), None, L._ptr(wt), dx.data_ptr(), M, N, K, N,
                       K, st)
            want_db = ctx.has_b and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                dw = _zeros(g, N, K)
                db = _zeros(g, N) if want_db else None  # bias gradient: summed inside the weight-gradient launch
                L.call("stcat_linear_wgrad", g.data_ptr(), x2.data_ptr(), dw.data_ptr(), L._ptr(db), M, N, K, N, K, st)
            elif want_db:
                db = colsum(g)
        else:
            dx = _empty(g, M, K) if ctx.needs_input_grad[0] else None
            dw = _empty(g, N, K) if ctx.needs_input_grad[1] else None
            db = _empty(g, N) if (ctx.has_b and dw is not None) else None
            L.call("stcat_small_linear_bwd", g.data_ptr(), x2.data_ptr(), w.data_ptr(), L._ptr(dx), L._ptr(dw),
                   L._ptr(db), M, N, K, st)
        dres = g.view(*ctx.xshape[:-1], N) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db, dres, None


def linear(x, w, b=None, res=None, relu=False):
    return LinearFn.apply(x, w, b, res, relu)


# ------------------------------------------------------------------------------------
# LayerNorm(x + res)
# ------------------------------------------------------------------------------------
class LayerNormFn(Function):
    """y = LayerNorm(res + dropout_p(x)) over 256 features (p = 0: plain x + res).  The residual branch's dropout
    (modal_encoder.py:237-240; query_decoder.py:344, 431, 436, 612, 653, 658) rides inside the kernel: mask from the
    counter stream in forward, regenerated in backward — no separate dropout pass over the activations."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, drop_p=0.0):
        shp = x.shape
        D = shp[-1]
        x2 = _c(x.reshape(-1, D))
        r2 = _c(res.reshape(-1, D)) if res is not None else None
        _chk(x2, r2, gamma, beta)
        M = x2.shape[0]
        y = torch.empty_like(x2)
        mean = _empty(x2, M)
        rstd = _empty(x2, M)
        drop = (0.0, 0, 0, None)
        if drop_p > 0.0:
            drop = (float(drop_p),) + _dropout_stream.take(x2.numel(), x2.device)
        L.call("stcat_layernorm_fwd", x2.data_ptr(), L._ptr(r2), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
               mean.data_ptr(), rstd.data_ptr(), M, D, eps, *drop, L.stream_of(x2))
        ctx.save_for_backward(x2, r2, gamma, mean, rstd)
        ctx.xshape = shp
        ctx.drop = drop
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, gamma, mean, rstd = ctx.saved_tensors
        M, D = x2.shape
        g = _c(dy.reshape(M, D))
        dz = torch.empty_like(x2)
        dx = torch.empty_like(x2) if ctx.drop[0] > 0.0 else None
        dgam = _zeros(x2, D)
        dbet = _zeros(x2, D)
        L.call("stcat_layernorm_bwd", g.data_ptr(), x2.data_ptr(), L._ptr(r2), gamma.data_ptr(), mean.data_ptr(),
               rstd.data_ptr(), dz.data_ptr(), L._ptr(dx), dgam.data_ptr(), dbet.data_ptr(), M, D, *ctx.drop,
               L.stream_of(g))
        dzv = dz.view(ctx.xshape)
        dxv = dx.view(ctx.xshape) if dx is not None else dzv
        return dxv, (dzv if r2 is not None else None), dgam, dbet, None, None


def layer_norm(x, gamma, beta, res=None, eps: float = 1e-5, drop_p: float = 0.0):
    return LayerNormFn.apply(x, res, gamma, beta, eps, drop_p)


# ------------------------------------------------------------------------------------
# element-wise glue with gradients
# ------------------------------------------------------------------------------------
class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        assert a.shape == b.shape
        return ew(L.EW_ADD, a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


class Add3Fn(Function):
    @staticmethod
    def forward(ctx, a, b, c):
        assert a.shape == b.shape == c.shape
        return ew(L.EW_ADD3, a, b, c)

    @staticmethod
    def backward(ctx, g):
        return g, g, g


class AddConstFn(Function):
    """a + c where c carries no gradient and may be a row-broadcast [D] / [rows,D] block."""

    @staticmethod
    def forward(ctx, a, c):
        assert a.numel() % c.numel() == 0
        return ew(L.EW_ADD, a, c, bmod=c.numel())

    @staticmethod
    def backward(ctx, g):
        return g, None


class MulFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        assert a.shape == b.shape
        ctx.save_for_backward(a, b)
        return ew(L.EW_MUL, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _c(g)
        return ew(L.EW_MUL, g, b), ew(L.EW_MUL, g, a)


class AffineRowsFn(Function):
    """x[M,D] * gamma[D] + beta[D]  (template generator FiLM, query_decoder.py:465-468)."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        M, D = x.shape
        ctx.save_for_backward(x, gamma)
        t = ew(L.EW_MUL, x, gamma, bmod=D)
        return ew(L.EW_ADD, t, beta, bmod=D, out=t)

    @staticmethod
    def backward(ctx, g):
        x, gamma = ctx.saved_tensors
        g = _c(g)
        D = x.shape[1]
        return ew(L.EW_MUL, g, gamma, bmod=D), colsum(g, _c(x)), colsum(g)


class UnaryFn(Function):
    @staticmethod
    def forward(ctx, x, fop, bop, save_out):
        y = ew(fop, x)
        ctx.bop = bop
        ctx.save_for_backward(y if save_out else _c(x))
        return y

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        return ew(ctx.bop, _c(g), s), None, None, None


def add(a, b):
    return AddFn.apply(a, b)


def add3(a, b, c):
    return Add3Fn.apply(a, b, c)


def add_const(a, c):
    return AddConstFn.apply(a, c)


def mul(a, b):
    return MulFn.apply(a, b)


def affine_rows(x, gamma, beta):
    return AffineRowsFn.apply(x, gamma, beta)


def sigmoid(x):
    return UnaryFn.apply(x, L.EW_SIGMOID, L.EW_SIGMOID_BWD, True)


def tanh(x):
    return UnaryFn.apply(x, L.EW_TANH, L.EW_TANH_BWD, True)


def inverse_sigmoid(x):
    """models/net_utils.py:59-63."""
    return UnaryFn.apply(x, L.EW_INVSIG, L.EW_INVSIG_BWD, False)


class SineEmbedFn(Function):
    """gen_sineembed_for_position — models/net_utils.py:29-56."""

    @staticmethod
    def forward(ctx, anchor):
        a2 = _c(anchor.reshape(-1, 4))
        _chk(a2)
        M = a2.shape[0]
        out = _empty(a2, M, 512)
        L.call("stcat_sine_embed_fwd", a2.data_ptr(), dim_t(a2.device).data_ptr(), out.data_ptr(), M, L.stream_of(a2))
        ctx.save_for_backward(a2)
        ctx.shp = anchor.shape
        return out.view(*anchor.shape[:-1], 512)

    @staticmethod
    def backward(ctx, g):
        (a2,) = ctx.saved_tensors
        M = a2.shape[0]
        g2 = _c(g.reshape(M, 512))
        da = torch.empty_like(a2)
        L.call("stcat_sine_embed_bwd", a2.data_ptr(), dim_t(a2.device).data_ptr(), g2.data_ptr(), da.data_ptr(), M,
               L.stream_of(a2))
        return da.view(ctx.shp)


def sine_embed(anchor):
    return SineEmbedFn.apply(anchor)


def pos_sine_2d(mask: torch.Tensor) -> torch.Tensor:
    """PositionEmbeddingSine(128, normalize=True): mask [n,h,w] bool -> [n, h*w, 256] (token-major)."""
    n, h, w = mask.shape
    m8 = _c(mask.to(torch.uint8))
    L.check_tensor(m8)
    pos = torch.empty(n, h * w, 256, device=mask.device, dtype=_f32)
    L.call("stcat_pos_sine_2d", m8.data_ptr(), dim_t(mask.device).data_ptr(), pos.data_ptr(), n, h, w,
           L.stream_of(pos))
    return pos


# ------------------------------------------------------------------------------------
# dropout (train mode)
# ------------------------------------------------------------------------------------
_MASK64 = (1 << 64) - 1


class _DropoutStream:
    """(seed, counter) of this process.  Every dropout site of a step takes a fresh counter range
    (csrc/stcat_rng.h); the backward pass replays the range saved by its forward.  The counter has two parts:
    a HOST offset that restarts at 0 every step (so a step always issues the same launch arguments) and a DEVICE
    base advanced by STEP_SPAN at `begin_step()` with a device-side add — which is what lets a captured hipGraph
    draw new masks on every replay.  DP ranks must seed differently (manual_seed(seed, rank)), exactly as torch's
    per-process generators differ."""

    STEP_SPAN = 1 << 36  # counters reserved per step (a C5 step uses ~2^33)
    SLOTS = 8            # device base words: a forward whose backward is still outstanding keeps its own

    def __init__(self):
        self.seed = None   # resolved lazily: manual_seed(), else from torch.initial_seed() and the DP rank
        self.offset = 0
        self._base = {}    # device -> int64 [1 + SLOTS]: word 0 = master counter, words 1.. = the slots the kernels read
        self._slot = {}    # device -> index of the current slot
        self.pending = False   # a train-mode forward ran and its backward has not finished yet
        self.trace = None      # debug / tests: a list that receives (host offset, decisions) of every site of the step

    def _default_seed(self) -> int:
        """Drop-in mode never calls manual_seed: the reference seeds torch with 42 + rank (scripts/train_net.py:296),
        so torch.initial_seed() already differs per rank; the rank is mixed in as well in case it does not."""
        rank = int(os.environ.get("RANK", "0"))
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
        except Exception:
            pass
        return _mix_seed(torch.initial_seed() & _MASK64, rank)

    def _words(self, device) -> torch.Tensor:
        b = self._base.get(device)
        if b is None:
            b = self._base[device] = torch.zeros(1 + self.SLOTS, dtype=torch.int64, device=device)
            self._slot[device] = 0
        return b

    def base(self, device) -> torch.Tensor:
        """the device word the kernels of the CURRENT forward (and of its backward) read their counter base from"""
        w = self._words(device)
        k = self._slot[device]
        return w[1 + k:2 + k]

    def take(self, numel: int, device):
        if self.seed is None:
            self.seed = self._default_seed()
        off = self.offset
        self.offset = off + ((numel + 3) // 4) * 4
        if self.trace is not None:
            self.trace.append((off, int(numel)))
        if self.offset > self.STEP_SPAN:
            raise RuntimeError("dropout counter range of one step exhausted (2^36 decisions in one forward pass)")
        return self.seed, off, self.base(device).data_ptr()

    def begin_step(self, device):
        """Open a new counter range.  The master word advances in place (stream-ordered, capturable: a replayed
        hipGraph draws fresh masks), then lands in the slot the next forward's kernels read.  The slot only moves on
        when the previous forward's backward is still outstanding (gradient accumulation over several clips, two
        losses): that backward regenerates its masks from ITS slot, which this step must not touch.  A plain
        forward / backward loop stays on one slot, so launch arguments are identical from step to step."""
        self.offset = 0
        w = self._words(device)
        capturing = w.is_cuda and torch.cuda.is_current_stream_capturing()
        if self.pending and not capturing:
            self._slot[device] = (self._slot[device] + 1) % self.SLOTS
        w[0:1].add_(self.STEP_SPAN)
        self.base(device).copy_(w[0:1])

    def auto_begin_step(self, device):
        """Called at the top of every train-mode forward (the visual encoder is the first module of the path to
        run): starts a new counter range iff decisions were drawn since the last begin_step, so an unmodified
        training loop (drop-in mode) never exhausts the per-step range, and an explicit dropout_begin_step()
        right before the forward is not doubled."""
        if self.offset != 0:
            self.begin_step(device)


_dropout_stream = _DropoutStream()


def _mix_seed(seed: int, rank: int) -> int:
    z = (seed * 0x9E3779B97F4A7C15 + (rank + 1) * 0xBF58476D1CE4E5B9) & _MASK64
    z ^= z >> 31
    return z & ((1 << 62) - 1)


def manual_seed(seed: int, rank: int = 0) -> None:
    """Seed the dropout stream of this process (counters restart at 0)."""
    _dropout_stream.seed = _mix_seed(seed, rank)
    _dropout_stream.offset = 0
    _dropout_stream.pending = False
    for d, b in _dropout_stream._base.items():
        b.zero_()
        _dropout_stream._slot[d] = 0


def dropout_begin_step(device) -> None:
    """Call once at the top of every training step (inside the captured region when the step is a hipGraph)."""
    _dropout_stream.begin_step(torch.device(device))


def dropout_auto_begin_step(device) -> None:
    """Start a new step's counter range if the previous one was used (see _DropoutStream.auto_begin_step)."""
    _dropout_stream.auto_begin_step(torch.device(device))


def dropout_forward_started() -> None:
    """a train-mode forward with gradients is under way: its masks belong to the current slot until its backward ran"""
    _dropout_stream.pending = True


def dropout_backward_done() -> None:
    """called by the last backward node of the path (encoder, then backbone): the slot may be reused by the next step"""
    _dropout_stream.pending = False
    if L.RECORDER is not None:
        L.RECORDER.effect(dropout_backward_done)


def dropout_trace(on: bool = True):
    """Test / debug aid: start (and return) the list of (host offset, decisions) pairs every dropout site appends to as it
    takes its counter range, in launch order — with `dropout_keep_mask(seed, base + offset, n, p)` a host program can
    rebuild the mask of every site of a step (tests/test_model_parity.py feeds them to the CPU oracle)."""
    _dropout_stream.trace = [] if on else None
    return _dropout_stream.trace


def dropout_stream_state():
    """(seed, host offset) — the device base is 0 until the first dropout_begin_step()"""
    if _dropout_stream.seed is None:
        _dropout_stream.seed = _dropout_stream._default_seed()
    return _dropout_stream.seed, _dropout_stream.offset


def dropout_keep_mask(seed: int, offset: int, n: int, p: float):
    """Host twin of csrc/stcat_rng.h (numpy uint64): bool[n], True = kept.  Test/debug helper; the product path
    never materialises a mask."""
    import numpy as np
    if p <= 0.0:
        return np.ones(n, dtype=bool)
    thresh = min(int(float(np.float32(p)) * 4294967296.0), 4294967295) or 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(offset) + np.uint64(1)
        z = np.uint64(seed) + ctr * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(32)) >= np.uint64(thresh)


class DropoutFn(Function):
    """y = res + dropout_p(x) (res optional): nn.Dropout of modal_encoder.py:237-240, query_decoder.py:344, 431-436,
    612, 653-658, net_utils.py:24-25.  Backward = the same launch on dY (mask regenerated, never stored)."""

    @staticmethod
    def forward(ctx, x, res, p):
        x = _c(x)
        r = _c(res) if res is not None else None
        seed, off, base = _dropout_stream.take(x.numel(), x.device)
        y = torch.empty_like(x)
        L.call("stcat_dropout", x.data_ptr(), L._ptr(r), y.data_ptr(), x.numel(), float(p), seed, off, base,
               L.stream_of(x))
        ctx.drop = (float(p), seed, off, base)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        dx = torch.empty_like(g)
        L.call("stcat_dropout", g.data_ptr(), None, dx.data_ptr(), g.numel(), *ctx.drop, L.stream_of(g))
        return dx, (g if ctx.has_res else None), None


def dropout(x, p: float):
    """dropout_p(x); callers pass p = 0 in eval mode (identity, no launch)."""
    return DropoutFn.apply(x, None, p) if p > 0.0 else x


def dropout_add(x, res, p: float):
    """res + dropout_p(x)."""
    return DropoutFn.apply(x, res, p) if p > 0.0 else add(x, res)


# ------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------
def _ld3(t: torch.Tensor) -> int:
    """row stride of a [B,S,D] view whose last dim is dense and whose batch stride is S*ld."""
    B, S, D = t.shape
    assert t.stride(2) == 1 and (B == 1 or t.stride(0) == S * t.stride(1)), "unsupported view"
    return t.stride(1)


# fp32-pipe self-attention: keep only (row max, 1 / row sum) and recompute the probability tiles in the backward instead of
# stashing the S x S matrix.  OPT-IN: it removes 515 MB of HBM traffic per encoder layer but costs 336 more 64-cycle fp32
# MFMAs per wave — measured +0.25 ms per C3 step (profiles/r05_as_kernel_experiments.log, item 8); it halves the
# attention's activation memory, which is what it is kept for.
MHA_RECOMPUTE = bool(os.environ.get("STCAT_MHA_RECOMPUTE"))
MHA_BS6_MIN_ROWS = int(os.environ.get("STCAT_MHA_BS6_MIN_ROWS", "128"))
MHA_FP32_PIPE = bool(os.environ.get("STCAT_MHA_FP32_PIPE"))      # A/B switch: every mode's self-attention on the fp32-pipe kernels


class MhaSelfFn(Function):
    """softmax(scale * q k^T + key_padding) v per (batch, head), head dim 32 — the core of
    torch.nn.MultiheadAttention after its in-projection (modal_encoder.py:236; query_decoder.py:341, 604).
    q, k, v: [B,S,256] views (e.g. column slices of a packed projection).  Returns (out, weights|None)."""

    @staticmethod
    def forward(ctx, q, k, v, kpm, scale, need_weights, packed_qk, drop_p=0.0):
        B, S, D = v.shape
        H = D // 32
        _chk(q, k, v)
        kp = None
        if kpm is not None:
            kp = _c(kpm.to(torch.uint8))
        SP = ((S + 31) // 32) * 32
        o = _empty(v, B, S, D)
        ctx.scale = scale
        ctx.packed_qk = packed_qk
        ctx.need_weights = need_weights
        # bf16-pipe kernels (csrc/attention_bs.h): online softmax, any S, only the row log-sum-exp is kept for backward.
        # The fp32-MFMA kernels remain for the exact-fp32 mode and for the one caller that consumes the head-mean
        # weights (the time decoder's self-attention, T queries).
        # (mode bf16x6p is fp32-class end to end: its attention runs on the fp32 matrix pipe as well)
        # rows longer than 256 tokens (non-square clips) train through the fp32 long-row kernels in every mode: the
        # bf16-pipe backward keeps a whole row's tiles in LDS and is built for S <= 256
        # (round 6: mode bf16x6p runs here too — three planes per operand, six products: csrc/attention_bs.h, NP = 3; the
        #  frozen experimental mode f16x3p keeps the fp32-pipe kernels)
        # In bf16x6p only rows longer than 128 tokens take the six-product kernels (the encoder's spatial layers: 207 at C3):
        # on the decoders' 64 queries and the temporal layers' 65 rows the fp32-pipe kernels are the faster ones — forward +
        # backward 36.0 vs 42.5 us at S = 64, 45.0 vs 48.5 at S = 65 (isolated, profiles/r06_attention.log: three planes of
        # 256 staged rows and 97 KB of LDS per workgroup do not pay on two or three tiles) — and those launches sit on the
        # step's latency-bound chains.
        mode = L.get_mma_mode()
        ctx.bs = ((not need_weights) and mode not in ("f32", "f16x3p") and not MHA_FP32_PIPE
                  and (mode != "bf16x6p" or S > MHA_BS6_MIN_ROWS)
                  and (S <= 256 or not any(ctx.needs_input_grad[:3])))
        if ctx.bs:
            keep = any(ctx.needs_input_grad[:3])
            lse = _empty(v, B, H, S) if keep else None
            drop = (0.0, 0, 0, None)
            if drop_p > 0.0:
                drop = (float(drop_p),) + _dropout_stream.take(B * H * SP * SP, v.device)
            L.call("stcat_mha_bs_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), L._ptr(kp), o.data_ptr(), L._ptr(lse),
                   B, H, S, _ld3(q), _ld3(k), _ld3(v), D, scale, *drop, L.stream_of(v))
            ctx.drop = drop
            if keep:
                ctx.save_for_backward(q, k, v, o, lse, kp)
            ctx.mark_non_differentiable()
            return o, None
        # fp32-pipe kernels.  When nobody reads the head-mean weights and a row fits the short-row kernels, the forward keeps
        # only (row maximum, 1 / row sum) and the backward recomputes its probability tiles (round 5): no S x S stash
        ctx.rc = (not need_weights) and S <= 256 and MHA_RECOMPUTE and any(ctx.needs_input_grad[:3])
        if ctx.rc:
            lse = _empty(v, B, H, SP, 2)
            drop = (0.0, 0, 0, None)
            if drop_p > 0.0:
                drop = (float(drop_p),) + _dropout_stream.take(B * H * SP * SP, v.device)
            L.call("stcat_mha_self_fwd_lse", q.data_ptr(), k.data_ptr(), v.data_ptr(), L._ptr(kp), o.data_ptr(), lse.data_ptr(),
                   B, H, S, _ld3(q), _ld3(k), _ld3(v), D, scale, *drop, L.stream_of(v))
            ctx.drop = drop
            ctx.save_for_backward(q, k, v, o, lse, kp)
            ctx.mark_non_differentiable()
            return o, None
        # probabilities are kept only when somebody will read them: backward, or the head-mean weights
        keep = need_weights or any(ctx.needs_input_grad[:3])
        pt = _empty(v, B, H, SP, SP) if keep else None
        drop = (0.0, 0, 0, None)
        if drop_p > 0.0:  # dropout on the probabilities (nn.MultiheadAttention(dropout=p) in train mode)
            drop = (float(drop_p),) + _dropout_stream.take(B * H * SP * SP, v.device)
        L.call("stcat_mha_self_fwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), L._ptr(kp), o.data_ptr(), L._ptr(pt),
               B, H, S, _ld3(q), _ld3(k), _ld3(v), D, scale, *drop, L.stream_of(v))
        wts = None
        if need_weights:
            wts = _empty(v, B, S, S)
            L.call("stcat_attn_weights_mean", pt.data_ptr(), wts.data_ptr(), B, H, S, *drop, L.stream_of(v))
        ctx.drop = drop
        if keep:
            ctx.save_for_backward(q, k, v, o, pt)
        if need_weights:
            return o, wts
        ctx.mark_non_differentiable()
        return o, None

    @staticmethod
    def backward(ctx, do, dwts):
        if ctx.bs:
            q, k, v, o, lse, kp = ctx.saved_tensors
            B, S, D = v.shape
            H = D // 32
            do = _c(do)
            if ctx.packed_qk:
                dqk = _empty(v, B, S, 2 * D)
                dq, dk, ldg_qk = dqk[:, :, :D], dqk[:, :, D:], 2 * D
            else:
                dq = _empty(v, B, S, D)
                dk = _empty(v, B, S, D)
                ldg_qk = D
            dv = _empty(v, B, S, D)
            L.call("stcat_mha_bs_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), L._ptr(kp), o.data_ptr(), do.data_ptr(),
                   lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, S, _ld3(q), _ld3(k), _ld3(v), D,
                   ldg_qk, D, ctx.scale, *ctx.drop, L.stream_of(v))
            if ctx.packed_qk:
                return dqk, None, dv, None, None, None, None, None
            return dq, dk, dv, None, None, None, None, None
        if getattr(ctx, "rc", False):
            q, k, v, o, lse, kp = ctx.saved_tensors
            B, S, D = v.shape
            H = D // 32
            do = _c(do)
            if ctx.packed_qk:
                dqk = _empty(v, B, S, 2 * D)
                dq, dk, ldg_qk = dqk[:, :, :D], dqk[:, :, D:], 2 * D
            else:
                dq = _empty(v, B, S, D)
                dk = _empty(v, B, S, D)
                ldg_qk = D
            dv = _empty(v, B, S, D)
            L.call("stcat_mha_self_bwd_lse", q.data_ptr(), k.data_ptr(), v.data_ptr(), L._ptr(kp), o.data_ptr(), do.data_ptr(),
                   lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, S, _ld3(q), _ld3(k), _ld3(v), D,
                   ldg_qk, D, ctx.scale, *ctx.drop, L.stream_of(v))
            if ctx.packed_qk:
                return dqk, None, dv, None, None, None, None, None
            return dq, dk, dv, None, None, None, None, None
        q, k, v, o, pt = ctx.saved_tensors
        B, S, D = v.shape
        H = D // 32
        SP = pt.shape[-1]
        do = _c(do)
        dw = corr = None
        if ctx.need_weights and dwts is not None:
            dw = _c(dwts)
            corr = _empty(v, B, H, S)
        dst = torch.empty_like(pt)
        if ctx.packed_qk:
            dqk = _empty(v, B, S, 2 * D)
            dq, dk, ldg_qk = dqk[:, :, :D], dqk[:, :, D:], 2 * D
        else:
            dq = _empty(v, B, S, D)
            dk = _empty(v, B, S, D)
            ldg_qk = D
        dv = _empty(v, B, S, D)
        L.call("stcat_mha_self_bwd", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(),
               pt.data_ptr(), L._ptr(dw), L._ptr(corr), dst.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
               B, H, S, _ld3(q), _ld3(k), _ld3(v), D, ldg_qk, D, ctx.scale, *ctx.drop, L.stream_of(v))
        if ctx.packed_qk:
            return dqk, None, dv, None, None, None, None, None
        return dq, dk, dv, None, None, None, None, None


def mha_self(q, k, v, kpm, scale, need_weights=False, drop_p=0.0):
    return MhaSelfFn.apply(q, k, v, kpm, scale, need_weights, False, drop_p)


def mha_self_packed(qk, v, kpm, scale, need_weights=False, drop_p=0.0):
    """q = qk[..., :D], k = qk[..., D:] come from one packed projection; the gradient is
    returned for the packed tensor directly."""
    D = v.shape[-1]
    return MhaSelfFn.apply(qk, qk[:, :, D:], v, kpm, scale, need_weights, True, drop_p)


class AttnQ1Fn(Function):
    """Time-aligned cross-attention, one query per frame (query_decoder.py:386-417 with the custom MHA
    of grounding_model/attention.py:184-393; query_decoder.py:618-639).
    q1/q2: [B,256] (q2 optional second 32-wide part per head); k1/k2: [B,S,256]; v: [B,S,256]."""

    @staticmethod
    def forward(ctx, q1, q2, k1, k2, v, kpm, scale, drop_p=0.0):
        B, S, D = v.shape
        H = D // 32
        q1, q2 = _c(q1), _c(q2)
        _chk(q1, q2, k1, k2, v)
        ldk, ldv = _ld3(k1), _ld3(v)  # k/v may be column slices of a wider (layer-batched) projection
        assert k2 is None or _ld3(k2) == ldk
        kp = _c(kpm.to(torch.uint8)) if kpm is not None else None
        out = _empty(v, B, D)
        P = _empty(v, B, H, S)
        drop = (0.0, 0, 0, None)
        if drop_p > 0.0:
            drop = (float(drop_p),) + _dropout_stream.take(B * H * S, v.device)
        ctx.drop = drop
        L.call("stcat_attn_q1_fwd", q1.data_ptr(), L._ptr(q2), k1.data_ptr(), L._ptr(k2), v.data_ptr(), L._ptr(kp),
               out.data_ptr(), P.data_ptr(), B, H, S, D, ldk, ldv, scale, *drop, L.stream_of(v))
        ctx.save_for_backward(q1, q2, k1, k2, v, P)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        q1, q2, k1, k2, v, P = ctx.saved_tensors
        B, S, D = v.shape
        H = D // 32
        g = _c(g)
        dq1 = torch.empty_like(q1)
        dq2 = torch.empty_like(q2) if q2 is not None else None
        dk1 = _empty(v, B, S, D)
        dk2 = _empty(v, B, S, D) if k2 is not None else None
        dv = _empty(v, B, S, D)
        L.call("stcat_attn_q1_bwd", q1.data_ptr(), L._ptr(q2), k1.data_ptr(), L._ptr(k2), v.data_ptr(), P.data_ptr(),
               g.data_ptr(), dq1.data_ptr(), L._ptr(dq2), dk1.data_ptr(), L._ptr(dk2), dv.data_ptr(), B, H, S, D,
               _ld3(k1), _ld3(v), ctx.scale, *ctx.drop, L.stream_of(v))
        return dq1, dq2, dk1, dk2, dv, None, None, None


class SplitColsFn(Function):
    """[M, n*D] -> n column blocks [M, D] (strided views).  Used to run the K/V projections of all decoder
    layers as ONE GEMM on the shared memory tensor; the backward concatenates the per-layer gradients so the
    batched GEMM also has a single dgrad / wgrad."""

    @staticmethod
    def forward(ctx, x, n):
        D = x.shape[-1] // n
        ctx.n, ctx.D = n, D
        ctx.like = x
        return tuple(x[..., i * D:(i + 1) * D] for i in range(n))

    @staticmethod
    def backward(ctx, *gs):
        x = ctx.like
        cols = [g if g is not None else torch.zeros(*x.shape[:-1], ctx.D, device=x.device, dtype=x.dtype) for g in gs]
        return torch.cat(cols, dim=-1), None


def split_cols(x, n):
    return SplitColsFn.apply(x, n)


class SplitRowsFn(Function):
    """Row blocks of a packed parameter (nn.MultiheadAttention's in_proj_weight [3D, D] / in_proj_bias [3D]) as
    views.  Plain slicing costs, per slice and step, a full-size zero fill + a copy in SliceBackward and an add to
    merge the slices' gradients (~500 launches per step over the 24 attention layers); here the backward is ONE
    concatenation of the per-block gradients."""

    @staticmethod
    def forward(ctx, x, sizes):
        assert sum(sizes) == x.shape[0]
        ctx.sizes = sizes
        ctx.rest = tuple(x.shape[1:])
        ctx.dev, ctx.dt = x.device, x.dtype
        out, a = [], 0
        for n in sizes:
            out.append(x[a:a + n])
            a += n
        return tuple(out)

    @staticmethod
    def backward(ctx, *gs):
        parts = [g if g is not None else torch.zeros(n, *ctx.rest, device=ctx.dev, dtype=ctx.dt)
                 for g, n in zip(gs, ctx.sizes)]
        return torch.cat(parts, dim=0), None


def split_rows(x, sizes):
    return SplitRowsFn.apply(x, tuple(sizes))


def attn_q1(q1, q2, k1, k2, v, kpm, scale, drop_p=0.0):
    return AttnQ1Fn.apply(q1, q2, k1, k2, v, kpm, scale, drop_p)


# ------------------------------------------------------------------------------------
# convolution building blocks (raw; composed by the backbone autograd function)
# ------------------------------------------------------------------------------------
def conv_out_hw(H, W, k, stride, pad):
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


def conv_fwd_raw(x, w_ohwi, scale, bias, res, stride, pad, relu):
    n, H, W, Cin = x.shape
    Cout, KH, KW, _ = w_ohwi.shape
    OH, OW = conv_out_hw(H, W, KH, stride, pad)
    y = _empty(x, n, OH, OW, Cout)
    L.call("stcat_conv_fwd", x.data_ptr(), w_ohwi.data_ptr(), L._ptr(scale), L._ptr(bias), L._ptr(res), y.data_ptr(),
           n, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), L.stream_of(x))
    return y


class WeightTransposer:
    """Transposed copies W[Cout,taps,Cin] -> [taps,Cin,Cout] of a fixed list of conv weights, refreshed with ONE
    launch per step (stcat_weight_transpose_multi) instead of one ~5 us launch per layer.  The transposed buffers
    and the device table are persistent; the table is rebuilt only if a weight moved."""

    def __init__(self):
        self.key = None
        self.table = None
        self.wts = {}
        self.total = 0

    def refresh(self, weights):
        """weights: list of OHWI tensors [Cout,KH,KW,Cin]; returns {weight.data_ptr(): transposed tensor}"""
        import numpy as np
        key = tuple(w.data_ptr() for w in weights)
        if key != self.key:
            if self.key is not None:      # a weight moved: launch plans hold the old table / the old raw pointers
                from . import plans
                plans.invalidate()
            dt = np.dtype([("w", "<u8"), ("wt", "<u8"), ("Cout", "<i4"), ("taps", "<i4"), ("Cin", "<i4"),
                           ("blk0", "<i4"), ("nbx", "<i4"), ("nby", "<i4")])
            assert dt.itemsize == L.load().stcat_weight_transpose_entry_bytes()
            tab = np.zeros(len(weights), dtype=dt)
            self.wts, blk = {}, 0
            for i, w in enumerate(weights):
                Cout, KH, KW, Cin = w.shape
                taps = KH * KW
                wt = torch.empty(taps, Cin, Cout, device=w.device, dtype=_f32)
                self.wts[w.data_ptr()] = wt
                nbx, nby = (Cin + 31) // 32, (Cout + 31) // 32
                tab[i] = (w.data_ptr(), wt.data_ptr(), Cout, taps, Cin, blk, nbx, nby)
                blk += nbx * nby * taps
            self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(weights[0].device)
            self.total, self.key, self.n = blk, key, len(weights)
        L.call("stcat_weight_transpose_multi", self.table.data_ptr(), self.n, self.total, L.stream_of(weights[0]))
        return self.wts


class LinearTransposes:
    """Transposed copies W^T [K, N] of the Linear weights whose data gradient runs on the transposed-operand kernel (rows
    M > 256: the encoder's projections / FFN, the decoders' memory-side projections — 59 of them per step, each a ~7 us
    launch on the backward pass's dependent chain before round 4).  Buffers are persistent; `refresh_all` rewrites every
    registered one with ONE launch (stcat_weight_transpose_multi) and is called by the first backward node of a step
    (StgLossFn.backward).  `get(w)` returns the copy when it is current — same parameter version, same WEIGHT_EPOCH — and
    otherwise transposes that one weight on the spot (a loop that never calls refresh_all, e.g. the reference's own
    criterion in drop-in mode, behaves as before)."""

    def __init__(self):
        self.entries = {}      # (data_ptr, N, K) -> [weakref to the weight's base tensor, transposed buffer, version, epoch]
        self.table = None
        self.total = 0
        self.dirty = True

    @staticmethod
    def _base(w: torch.Tensor) -> torch.Tensor:
        return w._base if w._base is not None else w

    def get(self, w: torch.Tensor) -> torch.Tensor:
        import weakref
        N, K = w.shape
        key = (w.data_ptr(), N, K)
        e = self.entries.get(key)
        if e is not None and e[0]() is None:          # the weight died and its address was handed out again
            e = None
        if e is not None and e[2] == w._version and e[3] == WEIGHT_EPOCH and e[1].device == w.device:
            # a cache hit while a launch plan is being RECORDED leaves the transpose launch out of the plan; a
            # replay is then only right if somebody refreshed this W^T earlier in the step (StgLossFn.backward does, a
            # foreign criterion or a second pass without an update does not).  The plan therefore carries a host-side
            # prerequisite: before every replay the entry is checked against the weight's version / WEIGHT_EPOCH and
            # re-transposed on the spot when it is stale.  (Under hipGraph capture there is no such hook: emit the launch.)
            if L.RECORDER is not None:
                L.RECORDER.prereq(lambda key=key: self._ensure(key))
                return e[1]
            if not (w.is_cuda and torch.cuda.is_current_stream_capturing()):
                return e[1]
        if e is None or e[1].device != w.device:
            # (a WEAK reference: the cache must not keep the parameters of every model a process ever built alive)
            e = self.entries[key] = [weakref.ref(self._base(w)), torch.empty(K, N, device=w.device, dtype=_f32), -1, -1]
            self.dirty = True
        L.call("stcat_weight_transpose", w.data_ptr(), e[1].data_ptr(), N, 1, K, L.stream_of(w))
        e[2], e[3] = w._version, WEIGHT_EPOCH
        return e[1]

    def _ensure(self, key) -> None:
        """replay-time check of one entry (see get): transpose now if the weight moved on since the entry was written"""
        e = self.entries.get(key)
        base = e[0]() if e is not None else None
        if base is None:
            return
        if e[2] != base._version or e[3] != WEIGHT_EPOCH:
            ptr, N, K = key
            L.call("stcat_weight_transpose", ptr, e[1].data_ptr(), N, 1, K, L.stream_of(base))
            e[2], e[3] = base._version, WEIGHT_EPOCH

    def refresh_all(self, like: torch.Tensor) -> None:
        import numpy as np
        dead = [k for k, e in self.entries.items() if e[0]() is None]
        for k in dead:
            del self.entries[k]
        if dead:
            self.dirty = True
        ents = [(k, e) for k, e in self.entries.items() if e[1].device == like.device]
        if not ents:
            return
        if self.dirty or self.table is None or self.table.device != like.device:
            if like.is_cuda and torch.cuda.is_current_stream_capturing():
                return      # (the table upload cannot be captured: the entries stay stale, get() transposes one by one)
            dt = np.dtype([("w", "<u8"), ("wt", "<u8"), ("Cout", "<i4"), ("taps", "<i4"), ("Cin", "<i4"),
                           ("blk0", "<i4"), ("nbx", "<i4"), ("nby", "<i4")])
            assert dt.itemsize == L.load().stcat_weight_transpose_entry_bytes()
            tab = np.zeros(len(ents), dtype=dt)
            blk = 0
            for i, ((ptr, N, K), e) in enumerate(ents):
                nbx, nby = (K + 31) // 32, (N + 31) // 32
                tab[i] = (ptr, e[1].data_ptr(), N, 1, K, blk, nbx, nby)
                blk += nbx * nby
            self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(like.device)
            self.total, self.n, self.dirty = blk, len(ents), False
        L.call("stcat_weight_transpose_multi", self.table.data_ptr(), self.n, self.total, L.stream_of(like))
        for _, e in ents:
            base = e[0]()
            if base is not None:
                e[2], e[3] = base._version, WEIGHT_EPOCH


LINEAR_WT = LinearTransposes()


def conv_dgrad_raw(g, w_ohwi, in_shape, stride, pad, add=None, out=None, mask_y=None, mask_scale=None,
                   scale2=None, wt=None):
    """mask_y / mask_scale: fuse the ReLU+FrozenBN backward of the layer that produced this conv's input.
    scale2: also return dx * scale2[c] as a second tensor (block-boundary form)."""
    n, H, W, Cin = in_shape
    Cout, KH, KW, _ = w_ohwi.shape
    dx = _empty(g, n, H, W, Cin) if out is None else out
    dx2 = _empty(g, n, H, W, Cin) if scale2 is not None else None
    if wt is None:
        wt = weight_transpose(w_ohwi.view(Cout, KH * KW, Cin))
    L.call("stcat_conv_dgrad", g.data_ptr(), w_ohwi.data_ptr(), L._ptr(add), L._ptr(mask_y), L._ptr(mask_scale),
           dx.data_ptr(), L._ptr(dx2), L._ptr(scale2), L._ptr(wt), n, H, W, Cin, Cout, KH, KW, stride, pad,
           L.stream_of(g))
    return dx if scale2 is None else (dx, dx2)


def weight_transpose(w3: torch.Tensor) -> torch.Tensor:
    """W [Cout, taps, Cin] -> [taps, Cin, Cout] (reduction-contiguous operand for the data-gradient GEMM)."""
    Cout, taps, Cin = w3.shape
    wt = _empty(w3, taps, Cin, Cout)
    L.call("stcat_weight_transpose", w3.data_ptr(), wt.data_ptr(), Cout, taps, Cin, L.stream_of(w3))
    return wt


def conv_wgrad_raw(g, x, w_shape_ohwi, stride, pad):
    n, H, W, Cin = x.shape
    Cout, KH, KW, _ = w_shape_ohwi
    dw = _zeros(g, Cout, KH, KW, Cin)
    L.call("stcat_conv_wgrad", g.data_ptr(), x.data_ptr(), dw.data_ptr(), n, H, W, Cin, Cout, KH, KW, stride, pad,
           L.stream_of(g))
    return dw


def stem_fwd_raw(frames, w_oihw, scale, bias):
    n, _, H, W = frames.shape
    OH, OW = conv_out_hw(H, W, 7, 2, 3)
    y = _empty(frames, n, OH, OW, 64)
    L.call("stcat_stem_fwd", frames.data_ptr(), w_oihw.data_ptr(), scale.data_ptr(), bias.data_ptr(), y.data_ptr(), n,
           H, W, L.stream_of(frames))
    return y


PIXEL_MEAN, PIXEL_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)  # cfg.INPUT.PIXEL_MEAN / PIXEL_STD of both experiment files
_U8_NORM = {}


def stem_u8_fwd_raw(frames_u8_hwc, w_oihw, scale, bias, mean=PIXEL_MEAN, std=PIXEL_STD):
    """uint8 [n,H,W,3] decoder frames -> stem output NHWC fp32; ToTensor + Normalize fused into the gather
    (datasets/vidstg.py:140, datasets/transforms.py:155-168)."""
    n, H, W, C = frames_u8_hwc.shape
    assert C == 3 and frames_u8_hwc.dtype == torch.uint8 and frames_u8_hwc.is_contiguous()
    L.check_tensor(frames_u8_hwc)
    key = (str(frames_u8_hwc.device), tuple(mean), tuple(std))
    if key not in _U8_NORM:
        m, s_ = torch.tensor(mean, dtype=torch.float64), torch.tensor(std, dtype=torch.float64)
        _U8_NORM[key] = ((1.0 / (255.0 * s_)).float().to(frames_u8_hwc.device), (-m / s_).float().to(frames_u8_hwc.device))
    isc, ish = _U8_NORM[key]
    OH, OW = conv_out_hw(H, W, 7, 2, 3)
    y = torch.empty(n, OH, OW, 64, device=frames_u8_hwc.device, dtype=_f32)
    L.call("stcat_stem_u8_fwd", frames_u8_hwc.data_ptr(), w_oihw.data_ptr(), isc.data_ptr(), ish.data_ptr(),
           scale.data_ptr(), bias.data_ptr(), y.data_ptr(), n, H, W, L.stream_of(frames_u8_hwc))
    return y


def maxpool_raw(x):
    n, H, W, C = x.shape
    OH, OW = conv_out_hw(H, W, 3, 2, 1)
    y = _empty(x, n, OH, OW, C)
    L.call("stcat_maxpool3x3s2", x.data_ptr(), y.data_ptr(), n, H, W, C, L.stream_of(x))
    return y


def frozen_bn_fold(w, b, rm, rv, eps: float = 1e-5):
    C = w.numel()
    scale = torch.empty_like(w)
    bias = torch.empty_like(w)
    L.call("stcat_frozen_bn_fold", w.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), scale.data_ptr(),
           bias.data_ptr(), C, eps, L.stream_of(w))
    return scale, bias


def temporal_map_argmax(pred_sted: torch.Tensor, durations: Sequence[int]) -> torch.Tensor:
    """PostProcess's live 2D temporal map (post_processor.py:30-53): returns int32 [b,2] (start, end)."""
    b, T, _ = pred_sted.shape
    sted = _c(pred_sted.detach())
    L.check_tensor(sted)
    dur = torch.tensor(list(durations), dtype=torch.int32, device=sted.device)
    out = torch.empty(b, 2, dtype=torch.int32, device=sted.device)
    L.call("stcat_temporal_map_argmax", sted.data_ptr(), dur.data_ptr(), out.data_ptr(), b, T, L.stream_of(sted))
    return out


# ------------------------------------------------------------------------------------
# plane-format backbone (mma mode "bf16x3p", csrc/igemm_pl.h): x = h + l as two bf16 planes
# ------------------------------------------------------------------------------------
_bf16 = torch.bfloat16
WEIGHT_EPOCH = 0  # bumped by the fused optimizer (it updates parameters through raw pointers: no torch version bump)


class Planes:
    """A tensor held as bf16 planes in ONE allocation [NP, *shape]: NP = 2 in mode bf16x3p (hi = bf16(x),
    lo = bf16(x - hi): 16 significand bits), NP = 3 in mode bf16x6p (hi + mid + lo == x exactly).  The C ABI takes the
    first two plane pointers; the third plane lies at the same spacing behind the second."""
    __slots__ = ("t", "mask")

    def __init__(self, t: torch.Tensor, mask: Optional[torch.Tensor] = None):
        # every plane dense; the planes equally spaced (a frame range of a whole-batch allocation is a valid plane set)
        assert t.dtype == _bf16 and t.shape[0] == L.plane_count() and t[0].is_contiguous(), (t.dtype, t.shape, L.get_mma_mode())
        self.t = t
        self.mask = mask  # optional bit mask (x > 0), uint8 [rows, C / 8], written by the producing conv epilogue

    def frames(self, a: int, b: int) -> "Planes":
        """frames a..b of an image plane set [NP, n, H, W, C] as a view (mask rows likewise)"""
        m = None
        if self.mask is not None:
            rows = self.t.shape[2] * self.t.shape[3]
            m = self.mask[a * rows:b * rows]
        return Planes(self.t[:, a:b], m)

    @staticmethod
    def empty(like: torch.Tensor, *shape) -> "Planes":
        return Planes(torch.empty((L.plane_count(), *shape), device=like.device, dtype=_bf16))

    @property
    def shape(self):
        return tuple(self.t.shape[1:])

    @property
    def device(self):
        return self.t.device

    def numel(self) -> int:
        return self.t[0].numel()

    @property
    def h(self) -> int:
        return self.t.data_ptr()

    @property
    def l(self) -> int:
        return self.t.data_ptr() + self.t.stride(0) * 2


def _pl(p: Optional[Planes]):
    return (None, None) if p is None else (p.h, p.l)


def pl_split(x: torch.Tensor) -> Planes:
    x = _c(x)
    _chk(x)
    out = Planes.empty(x, *x.shape)
    L.call("stcat_pl_split", x.data_ptr(), out.h, out.l, x.numel(), L.stream_of(x))
    return out


def pl_split_sum(x: torch.Tensor, y: torch.Tensor, want_sum: bool = True):
    """(x + y as fp32 | None, planes of x + y) in one pass"""
    x, y = _c(x), _c(y)
    _chk(x, y)
    assert x.shape == y.shape
    out = Planes.empty(x, *x.shape)
    sm = torch.empty_like(x) if want_sum else None
    L.call("stcat_pl_split_sum", x.data_ptr(), y.data_ptr(), L._ptr(sm), out.h, out.l, x.numel(), L.stream_of(x))
    return sm, out


def pl_join(p: Planes) -> torch.Tensor:
    out = torch.empty(p.shape, device=p.device, dtype=_f32)
    L.call("stcat_pl_join", p.h, p.l, out.data_ptr(), p.numel(), L.stream_of(out))
    return out


def pl_colsum(p: Planes) -> torch.Tensor:
    """column sums of a 2-D plane set [M, N] -> fp32 [N] (a bias gradient whose upstream gradient only exists as planes)"""
    M, N = p.shape
    out = _zeros(p.t, N)
    L.call("stcat_pl_colsum", p.h, p.l, out.data_ptr(), M, N, L.stream_of(p.t))
    return out


def pl_maxpool_raw(x: torch.Tensor) -> Planes:
    n, H, W, C = x.shape
    OH, OW = conv_out_hw(H, W, 3, 2, 1)
    y = Planes.empty(x, n, OH, OW, C)
    L.call("stcat_pl_maxpool3x3s2", x.data_ptr(), y.h, y.l, n, H, W, C, L.stream_of(x))
    return y


def pl_conv_fwd_raw(x: Planes, w: Planes, scale, bias, res: Optional[Planes], stride, pad, relu, planes_out=True,
                    f32_out=False, want_mask=False, out=None):
    """w: planes of the OHWI weight [Cout,KH,KW,Cin].  Returns (y planes | None, y fp32 | None); with want_mask the
    planes carry the bit mask (y > 0) for the backward pass.  out = (planes | None, fp32 | None): write there (a frame
    range of a whole-batch allocation) instead of allocating."""
    n, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH, OW = conv_out_hw(H, W, KH, stride, pad)
    if out is not None:
        yp, yf = out
    else:
        yp = Planes.empty(x.t, n, OH, OW, Cout) if planes_out else None
        yf = torch.empty(n, OH, OW, Cout, device=x.device, dtype=_f32) if f32_out else None
        if want_mask and yp is not None:
            yp.mask = torch.empty(n * OH * OW, Cout // 8, device=x.device, dtype=torch.uint8)
    L.call("stcat_pl_conv_fwd", x.h, x.l, w.h, w.l, L._ptr(scale), L._ptr(bias), *_pl(res), *_pl(yp), L._ptr(yf),
           L._ptr(yp.mask if yp is not None else None), n, H, W, Cin, Cout, KH, KW, stride, pad, int(relu),
           L.stream_of(x.t))
    return yp, yf


def pl_conv_dgrad_raw(g: Planes, wt: Planes, in_shape, k, stride, pad, add: Optional[Planes] = None,
                      out: Optional[Planes] = None, mask_y: Optional[Planes] = None, mask_scale=None, scale2=None):
    """wt: TRANSPOSED weight planes [taps, Cin, Cout].  Semantics of conv_dgrad_raw on planes."""
    n, H, W, Cin = in_shape
    Cout = g.shape[-1]
    dx = Planes.empty(g.t, n, H, W, Cin) if out is None else out
    dx2 = Planes.empty(g.t, n, H, W, Cin) if scale2 is not None else None
    bits = mask_y.mask if mask_y is not None else None
    L.call("stcat_pl_conv_dgrad", g.h, g.l, wt.h, wt.l, *_pl(add), *(_pl(mask_y) if bits is None else (None, None)),
           L._ptr(bits), L._ptr(mask_scale), dx.h, dx.l,
           *_pl(dx2), L._ptr(scale2), n, H, W, Cin, Cout, k, k, stride, pad, L.stream_of(g.t))
    return dx if scale2 is None else (dx, dx2)


def pl_conv_dgrad_cadd_raw(g: Planes, wt: Planes, in_shape, addc: Planes, add_stride: int, mask_y: Optional[Planes] = None,
                           mask_scale=None, out: Optional[Planes] = None) -> Planes:
    """1x1 stride-1 data gradient + the COARSE-grid operand `addc` [n, ceil(H/s), ceil(W/s), Cin] scattered onto the
    stride-s lattice (the downsample branch's gradient, never materialised at full resolution) + bit-mask ReLU backward"""
    n, H, W, Cin = in_shape
    Cout = g.shape[-1]
    dx = Planes.empty(g.t, n, H, W, Cin) if out is None else out
    bits = mask_y.mask if mask_y is not None else None
    assert mask_y is None or bits is not None, "the coarse-add form takes the ReLU mask as bits"
    L.call("stcat_pl_conv_dgrad_cadd", g.h, g.l, wt.h, wt.l, addc.h, addc.l, int(add_stride), L._ptr(bits),
           L._ptr(mask_scale), dx.h, dx.l, n, H, W, Cin, Cout, L.stream_of(g.t))
    return dx


def pl_conv_wgrad_raw(g: Planes, x: Planes, w_shape_ohwi, stride, pad, row_scale=None, out=None) -> torch.Tensor:
    """row_scale [Cout]: dW[co] *= row_scale[co] — a FrozenBN scale folded out of g (g = dz, not dz * scale).
    out: a ZEROED OHWI buffer to accumulate into (a data-parallel gradient bucket) instead of a fresh one."""
    n, H, W, Cin = x.shape
    Cout, KH, KW, _ = w_shape_ohwi
    dw = _zeros(g.t, Cout, KH, KW, Cin) if out is None else out
    assert dw.is_contiguous() and tuple(dw.shape) == (Cout, KH, KW, Cin)
    st = L.stream_of(g.t)
    ws = _wgrad_workspace(g.t.device, st)
    if ws is not None:
        L.call("stcat_pl_conv_wgrad_ws", g.h, g.l, x.h, x.l, dw.data_ptr(), L._ptr(row_scale), n, H, W, Cin, Cout, KH, KW,
               stride, pad, ws.data_ptr(), ws.numel(), st)
    else:
        L.call("stcat_pl_conv_wgrad", g.h, g.l, x.h, x.l, dw.data_ptr(), L._ptr(row_scale), n, H, W, Cin, Cout, KH, KW,
               stride, pad, st)
    return dw


# Workspace of the atomics-free plane weight gradient (csrc/igemm_pl.h: pl_wgrad_reduce_kernel): one buffer per (device,
# stream) — launches of one stream use it one after the other.  96 MB covers every layer of the backbone at C3 (the largest:
# 28 slices x 256 x 2304 values = 66 MB) and the encoder FFN; a launch that needs more falls back to atomics.  Created by
# the first EAGER call on a stream (a launch-plan recording cannot allocate persistent memory; its eager pass comes first).
WGRAD_WS_FLOATS = int(os.environ.get("STCAT_WGRAD_WS_MB", "96")) * (1 << 18)
_WGRAD_WS = {}


def _wgrad_workspace(device, stream):
    if WGRAD_WS_FLOATS <= 0 or device.type != "cuda":
        key = (str(device), 0)
    else:
        key = (str(device), int(stream) if stream is not None else 0)
    ws = _WGRAD_WS.get(key)
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if ws is None and WGRAD_WS_FLOATS > 0:
        if L.RECORDER is not None or capturing:
            return None       # (a recording / a hipGraph capture cannot allocate persistent memory: atomics path)
        ws = _WGRAD_WS[key] = torch.empty(WGRAD_WS_FLOATS if device.type == "cuda" else min(WGRAD_WS_FLOATS, 1 << 22),
                                          device=device, dtype=_f32)
    rec = L.RECORDER
    if ws is not None and rec is not None and device.type == "cuda" and rec.slots.get(key[1]) == 0:
        #  the workspace is keyed by the stream the launch was RECORDED on, but a replay maps slot 0 to
        # whatever stream is current then: replayed elsewhere, the plan would share this buffer with eager launches of
        # the recorded stream, unordered.  Refuse such a replay instead of racing.
        want = key[1]

        def _same_stream(want=want, device=device):
            if torch.cuda.current_stream(device).cuda_stream != want:
                raise L.StcatHipError("launch plan: recorded with a weight-gradient workspace of another stream; replay "
                                      "it on the stream it was recorded on, or call stcat_amd.plans.clear()")
        rec.prereq(_same_stream)
    return ws


def free_wgrad_workspaces() -> None:
    """drop the per-stream workspaces (plans.clear(): the plans that baked their addresses are gone)"""
    _WGRAD_WS.clear()


def pl_act_bwd_raw(dy: torch.Tensor, y: Optional[torch.Tensor], scale, want_g=True, want_res=False, relu=True):
    dy = _c(dy)
    C = dy.shape[-1]
    G = Planes.empty(dy, *dy.shape) if want_g else None
    R = Planes.empty(dy, *dy.shape) if want_res else None
    L.call("stcat_pl_act_bwd", dy.data_ptr(), L._ptr(y), L._ptr(scale), *_pl(G), *_pl(R), dy.numel(), C, int(relu),
           L.stream_of(dy))
    return G, R


def pl_scale_raw(x: Planes, scale: torch.Tensor) -> Planes:
    out = Planes.empty(x.t, *x.shape)
    L.call("stcat_pl_scale", x.h, x.l, scale.data_ptr(), out.h, out.l, x.numel(), x.shape[-1], L.stream_of(x.t))
    return out


class WeightPlanes:
    """bf16 hi/lo planes of a fixed list of conv weights, refreshed with ONE launch: forward planes [Cout,KH,KW,Cin]
    and (when `transposed`) the data-gradient operand [taps,Cin,Cout].  Buffers and the device table are persistent;
    a refresh is skipped when no weight changed since the last one (tensor versions + the optimizer epoch)."""

    def __init__(self):
        self.key = None
        self.state = None
        self.table = None
        self.fwd = {}
        self.tr = {}
        self.total = 0
        self.n = 0

    def refresh(self, weights, transposed: bool, tscales=None):
        """weights: list of OHWI fp32 tensors; tscales: optional list (one entry per weight, None = 1) of [Cout] factors
        folded into the TRANSPOSED planes; returns ({ptr: Planes fwd}, {ptr: Planes transposed})"""
        import numpy as np
        tscales = tscales if tscales is not None else [None] * len(weights)
        # (the plane count is part of the key — ADVICE r03: switching bf16x3p -> bf16x6p on a live model must not reuse the
        # two-plane buffers, the three-plane kernels would write / read a third plane past their end)
        want_tr = bool(transposed) or bool(self.tr)
        key = (tuple(w.data_ptr() for w in weights), want_tr,
               tuple(0 if t is None else t.data_ptr() for t in tscales), L.plane_count(), L.get_mma_mode())
        if key != self.key:
            if self.key is not None:      # a weight moved / the plane mode changed: launch plans hold the old table
                from . import plans
                plans.invalidate()
            dt = np.dtype([("w", "<u8"), ("wh", "<u8"), ("wl", "<u8"), ("th", "<u8"), ("tl", "<u8"), ("tscale", "<u8"),
                           ("Cout", "<i4"), ("taps", "<i4"), ("Cin", "<i4"), ("blk0", "<i4"), ("nbx", "<i4"),
                           ("nby", "<i4"), ("pad", "<i4"), ("pad2", "<i4")])
            assert dt.itemsize == L.load().stcat_weight_planes_entry_bytes(), dt.itemsize
            tab = np.zeros(len(weights), dtype=dt)
            self.fwd, self.tr, blk = {}, {}, 0
            for i, w in enumerate(weights):
                Cout, KH, KW, Cin = w.shape
                taps = KH * KW
                wp = Planes.empty(w, Cout, KH, KW, Cin)
                self.fwd[w.data_ptr()] = wp
                th = tl = 0
                if want_tr:
                    tp = Planes.empty(w, taps, Cin, Cout)
                    self.tr[w.data_ptr()] = tp
                    th, tl = tp.h, tp.l
                nbx, nby = (Cin + 31) // 32, (Cout + 31) // 32
                ts = tscales[i].data_ptr() if (want_tr and tscales[i] is not None) else 0
                tab[i] = (w.data_ptr(), wp.h, wp.l, th, tl, ts, Cout, taps, Cin, blk, nbx, nby, 0, 0)
                blk += nbx * nby * taps
            self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(weights[0].device)
            self.total, self.key, self.n, self.state = blk, key, len(weights), None
        state = (WEIGHT_EPOCH, tuple(w._version for w in weights))
        # (while a hipGraph is being captured the launch must be part of it: the replayed step follows an optimizer
        # step that moved the fp32 weights — ADVICE r02)
        capturing = (weights[0].is_cuda and torch.cuda.is_current_stream_capturing()) or L.RECORDER is not None
        if state != self.state or capturing:
            np_ = L.plane_count()
            assert all(p.t.shape[0] == np_ for p in self.fwd.values()) and all(p.t.shape[0] == np_ for p in self.tr.values())
            L.call("stcat_weight_planes_multi", self.table.data_ptr(), self.n, self.total, L.stream_of(weights[0]))
            self.state = state
        return self.fwd, self.tr


_WGRAD_STREAMS = {}
WGRAD_STREAM_ENABLED = not os.environ.get("STCAT_NO_WGRAD_STREAM")


class single_stream:
    """`with ops.single_stream():` — every launch of the step on the caller's stream (no forked decoder, no second
    forward chain, no weight-gradient stream): kernels run one at a time, so a per-launch duration measured inside is
    the kernel's ISOLATED duration (bench.py's serialised instrumented step).  Launch plans carry both
    switches in their signature, so eager / recorded steps inside and outside the block never share a plan."""

    def __enter__(self):
        global FORK_ENABLED, WGRAD_STREAM_ENABLED
        self.saved = (FORK_ENABLED, WGRAD_STREAM_ENABLED)
        FORK_ENABLED = WGRAD_STREAM_ENABLED = False
        return self

    def __exit__(self, *exc):
        global FORK_ENABLED, WGRAD_STREAM_ENABLED
        FORK_ENABLED, WGRAD_STREAM_ENABLED = self.saved
        return False


class WgradStream:
    """`with WgradStream(x): <launch>` puts a weight-gradient launch on a second HIP stream, ordered after everything
    queued so far on the current stream; `keep(...)` tells the caching allocator which tensors that launch reads;
    `join(*outputs)` makes the current stream wait for all of them.  A no-op on the emulator / when disabled."""

    def __init__(self, like: torch.Tensor):
        self.active = WGRAD_STREAM_ENABLED and L._backend == "hip" and like.is_cuda
        if self.active:
            dev = like.device
            self.main = torch.cuda.current_stream(dev)
            self.side = _WGRAD_STREAMS.get(dev)
            if self.side is None:
                self.side = _WGRAD_STREAMS[dev] = side_stream(dev, 1)
            self.ctx = None

    def __enter__(self):
        if self.active:
            _wait_stream(self.side, self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.ctx.__exit__(*exc)
        return False

    def keep(self, *tensors):
        if self.active:
            for t in tensors:
                t = t.t if isinstance(t, Planes) else t
                if torch.is_tensor(t):
                    t.record_stream(self.side)
                    # A launch plan bakes the allocator's decisions of the recording pass into addresses: a block
                    # the side stream reads must not be handed out again later in the same sequence (at replay the
                    # side stream may lag behind where it was when the allocator judged the block free)
                    if L.RECORDER is not None:
                        L.RECORDER.keep.append(t)

    def join(self, *outputs):
        if self.active:
            _wait_stream(self.main, self.side)
            for t in outputs:
                if torch.is_tensor(t):
                    t.record_stream(self.main)
