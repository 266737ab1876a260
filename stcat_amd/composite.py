"""Composite autograd functions: ONE `torch.autograd.Function` per transformer layer of the grounding model.

Round-1/2 profile (profiles/r02_host_profile.log): the grounding model is ~530 autograd nodes of one or
two kernels each; `Function.apply` costs ~12 us per node forward and ~45 us through the backward engine, gradient sums
of multi-consumer tensors run as `at::native` add kernels and packed-parameter slices come back through `cat`.  Here a
layer's forward is a straight sequence of kernel launches and its backward is written out by hand in reverse order:
one autograd node per layer, gradient sums fused into the data-gradient epilogues (`add=`) or our own element-wise
kernel, packed in-projection gradients written straight into their row blocks.  The kernels (and the reference call
sites they replace) are the ones of `stcat_amd.ops`; only the host-side wiring differs, so every parity test of the
model covers this path.  `STCAT_NO_COMPOSITE=1` switches back to the op-by-op wiring.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch.autograd import Function

from . import _lib as L
from . import ops, plans

ENABLED = not os.environ.get("STCAT_NO_COMPOSITE")


class _Ctx:
    """the subset of the autograd ctx API that the Functions of stcat_amd.ops use"""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def mark_non_differentiable(self, *a):
        pass


def _f(fn, nig, *args):
    """run fn.forward outside autograd; returns (outputs, ctx)"""
    c = _Ctx(nig)
    return fn.forward(c, *args), c


_T = (True,) * 8


def _add(a, b):
    return ops.ew(L.EW_ADD, a, b)


def _lin_f(x, w, b, res=None, relu=False):
    shp = x.shape
    x2 = x if x.dim() == 2 else x.reshape(-1, shp[-1])
    if not (x2.is_contiguous() or (x2.stride(1) == 1 and w.shape[0] % 64 == 0)):      # row-strided is fine for the GEMM
        x2 = x2.contiguous()
    r2 = res.reshape(-1, w.shape[0]) if res is not None else None
    if r2 is not None and not r2.is_contiguous():
        r2 = r2.contiguous()
    y = ops.linear_fwd_raw(x2, w, b, r2, relu)
    return y.view(*shp[:-1], w.shape[0]), x2


def _lin_b(g, x2, w, need_dx=True, dw=None, db=None, add=None, relu_y=None, want_db=True, mask=None):
    """backward of y = relu?(x2 w^T + b (+ res)): returns (dx [M,K] | None, dw, db, g_after_relu).  `dw` / `db`: zeroed
    buffers to accumulate into (row blocks of a packed parameter's gradient); `add`: [M,K] tensor summed into dx inside
    the data-gradient launch.  `mask` = (y [M,K], gain): the Linear's INPUT was y = dropout(relu(.)) — its ReLU and
    dropout backward ride in the data gradient's epilogue (dx = [y > 0] * gain * (g w))."""
    N, K = w.shape
    g = g.reshape(-1, N)
    g = g if g.is_contiguous() else g.contiguous()
    if relu_y is not None:
        g, _ = ops.act_bwd_raw(g, relu_y.reshape(-1, N), None, want_g=True, relu=True)
    M = g.shape[0]
    st = L.stream_of(g)
    dx = None
    if N % 64 == 0:
        if need_dx:
            if add is not None:
                add = add.reshape(M, K)
                add = add if add.is_contiguous() else add.contiguous()
            dx = (ops._zero_take(g, (M, K)) if (M <= 128 and N >= 128 and L.get_mma_mode() != "f32" and mask is None)
                  else None)
            if mask is not None and mask[0] == "bits":
                # the Linear's input was dropout(relu(.)) written by the plane kernel with its bit mask: data gradient on the
                # A-stationary plane kernel (K_red = N = 256 -> 2048 columns), ReLU + dropout backward from the bits
                _, ybits, gainvec, wtp = mask
                gp = ops.pl_split(g)
                dx = torch.empty(M, K, device=g.device, dtype=torch.float32)
                L.call("stcat_pl_linear_dgrad_mask", gp.h, gp.l, wtp.h, wtp.l, ybits.data_ptr(), gainvec.data_ptr(),
                       dx.data_ptr(), None, None, M, N, K, st)
            elif mask is not None:
                my, gain = mask
                dx = torch.empty(M, K, device=g.device, dtype=torch.float32)
                wt = ops.LINEAR_WT.get(w) if M > 256 else None
                L.call("stcat_linear_dgrad_mask", g.data_ptr(), w.data_ptr(), L._ptr(add), L._ptr(wt), my.data_ptr(),
                       float(gain), dx.data_ptr(), M, N, K, N, K, st)
            elif dx is not None:       # skinny: zeroed output, reduction split over grid.z (see ops.linear_fwd_raw)
                L.call("stcat_linear_dgrad_acc", g.data_ptr(), w.data_ptr(), L._ptr(add), dx.data_ptr(), M, N, K, N, K, st)
            else:
                dx = torch.empty(M, K, device=g.device, dtype=torch.float32)
                wt = ops.LINEAR_WT.get(w) if M > 256 else None
                L.call("stcat_linear_dgrad", g.data_ptr(), w.data_ptr(), L._ptr(add), L._ptr(wt), dx.data_ptr(), M, N, K, N, K, st)
        if dw is None:
            dw = ops._zeros(g, N, K)
        if db is None and want_db:
            db = ops._zeros(g, N)
        _wgrad(g, x2, dw, db, M, N, K)
    else:
        #  the 4 / 2 / 1-wide heads: this kernel has no epilogue for a fused ReLU + dropout backward
        assert mask is None, "_lin_b: mask= needs the 64-multiple data-gradient kernels (N % 64 == 0)"
        dx_ = torch.empty(M, K, device=g.device, dtype=torch.float32) if need_dx else None
        dw_ = torch.empty(N, K, device=g.device, dtype=torch.float32)
        db_ = torch.empty(N, device=g.device, dtype=torch.float32) if want_db else None
        L.call("stcat_small_linear_bwd", g.data_ptr(), x2.data_ptr(), w.data_ptr(), L._ptr(dx_), dw_.data_ptr(),
               L._ptr(db_), M, N, K, st)
        dx = dx_ if add is None or dx_ is None else _add(dx_, add.reshape(M, K))
        dw = dw_ if dw is None else ops.ew(L.EW_ADD, dw, dw_, out=dw)
        db = db_ if db is None else (ops.ew(L.EW_ADD, db, db_, out=db) if db_ is not None else db)
    return dx, dw, db, g


# ------------------------------------------------------------------------------------------------------------------
# weight gradients leave the critical path
# ------------------------------------------------------------------------------------------------------------------
# A layer's backward is a dependent chain (data gradient -> LayerNorm backward -> attention backward -> ...) of small
# launches; its weight / bias gradients feed nothing but the optimizer.  Inside a `wgrad_batch` they are collected and
# issued together on the weight-gradient stream (ops.WgradStream, shared with the backbone) when the layer's chain has
# been queued: the chain no longer waits behind ~270 weight-gradient launches per step (5.4 ms of kernel time at C3,
# profiles/r03_timeline.log), which run on CUs the chain leaves idle.  The outermost batch joins the streams, so a node's
# backward returns with every gradient ordered on its own stream, as autograd expects.
_BATCH = []          # stack of pending launch lists (innermost last)
_DIRTY = [False]     # the side stream holds launches the current stream has not waited for yet


DEFER_WGRADS = not os.environ.get("STCAT_NO_WGRAD_DEFER")


INLINE_TIME_WGRADS = not os.environ.get("STCAT_TIME_WGRADS_DEFERRED")
PROJ_LANE = not os.environ.get("STCAT_NO_PROJ_LANE")


def _wgrad(g, x2, dw, db, M, N, K):
    args = (g.data_ptr(), x2.data_ptr(), dw.data_ptr(), L._ptr(db), M, N, K, N, x2.stride(0))
    if _BATCH and _BATCH[-1] is not None and DEFER_WGRADS:
        _BATCH[-1].append((args, g, x2))
    else:
        L.call("stcat_linear_wgrad", *args, L.stream_of(g))


def _wgrad_multi(gs, xs, dws, dbs, M, N, K):
    """the weight gradients of up to eight same-shape skinny Linears as ONE launch (ops.linear_wgrad_multi), deferred to the
    weight-gradient stream like _wgrad"""
    if _BATCH and _BATCH[-1] is not None and DEFER_WGRADS:
        _BATCH[-1].append((("multi", list(gs), list(xs), list(dws), list(dbs), M, N, K), gs[0], xs[0]))
    else:
        ops.linear_wgrad_multi(gs, xs, dws, dbs, M, N, K)


def wgrad_flush(like: torch.Tensor) -> None:
    """issue what the innermost batch has collected so far (end of a decoder layer inside its node's batch)"""
    if not _BATCH or not _BATCH[-1]:       # (no batch, an empty one, or a None frame: launches went out directly)
        return
    pending = _BATCH[-1]
    _BATCH[-1] = []
    wg = ops.WgradStream(like)
    with wg:
        st = L.stream_of(like)
        for args, _, _ in pending:
            if args[0] == "multi":
                ops.linear_wgrad_multi(*args[1:])
            elif args[0] == "call":
                args[1]()
            else:
                L.call("stcat_linear_wgrad", *args, st)
    for args, g, x2 in pending:
        if args[0] == "multi":
            wg.keep(*args[1], *args[2])
        elif args[0] == "call":
            wg.keep(*args[2])
        else:
            wg.keep(g, x2)
    _DIRTY[0] = True


class wgrad_batch:
    """`with wgrad_batch(like): <a layer's backward>`; batches nest (a node's backward around its layers)"""

    def __init__(self, like: torch.Tensor):
        self.like = like

    def __enter__(self):
        _BATCH.append([])
        return self

    def __exit__(self, et, ev, tb):
        if et is not None:
            _BATCH.pop()
            return False
        wgrad_flush(self.like)
        _BATCH.pop()
        if not _BATCH and _DIRTY[0]:
            ops.WgradStream(self.like).join()
            _DIRTY[0] = False
        return False


class _Lane:
    """a second launch lane beside a node's dependent chain (one of the package's measured-concurrent side streams):
    `fork()` orders it behind everything queued on the current stream, `with lane:` queues launches on it, `sync_main()`
    makes the current stream wait for what the lane holds SO FAR (launches queued on the lane afterwards run beside the
    chain), `keep(...)` tells the allocator / the recording plan that the lane touches these tensors."""

    def __init__(self, like: torch.Tensor, index: int, enabled: bool = True):
        self.active = bool(enabled and ops.FORK_ENABLED and L._backend == "hip" and like.is_cuda)
        if self.active:
            self.main = torch.cuda.current_stream(like.device)
            self.side = ops.side_stream(like.device, index)
            self.active = self.side.cuda_stream != self.main.cuda_stream
        self.ctx = None

    def fork(self):
        if self.active:
            ops._wait_stream(self.side, self.main)

    def sync_main(self):
        if self.active:
            ops._wait_stream(self.main, self.side)

    def __enter__(self):
        if self.active:
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
                if torch.is_tensor(t):
                    t.record_stream(self.side)
                    if L.RECORDER is not None:
                        L.RECORDER.keep.append(t)


# ------------------------------------------------------------------------------------------------------------------
# shared sub-blocks
# ------------------------------------------------------------------------------------------------------------------
def _outln_f(a, Wo, bo, res, g, be, p):
    """LayerNorm(res + dropout_p(a Wo^T + bo))  (modal_encoder.py:237-238, query_decoder.py:343-345, 431-432, 611-613, 653-654)"""
    if p > 0.0:
        h, x_o = _lin_f(a, Wo, bo)
        y, c_n = _f(ops.LayerNormFn, _T, h, res, g, be, 1e-5, p)
    else:
        h, x_o = _lin_f(a, Wo, bo, res=res)          # eval: the residual rides in the GEMM epilogue
        y, c_n = _f(ops.LayerNormFn, _T, h, None, g, be, 1e-5, 0.0)
    return y, (c_n, x_o, Wo, p, y.shape)


def _outln_b(st, dy, need_da=True, mask=None):
    """-> (d_a [M,K], d_res (shape of y), dWo, dbo, dg, dbe); mask: see _lin_b (the FFN's fused ReLU + dropout backward)"""
    c_n, x_o, Wo, p, shp = st
    r = ops.LayerNormFn.backward(c_n, dy.reshape(shp))
    d_h, d_res, dg, dbe = r[0], r[1], r[2], r[3]
    if p == 0.0:
        d_res = d_h
    d_a, dWo, dbo, _ = _lin_b(d_h, x_o, Wo, need_dx=need_da, mask=mask)
    return d_a, d_res, dWo, dbo, dg, dbe


FUSE_FFN = not os.environ.get("STCAT_NO_FFN_FUSE")


FFN_PLANES = not os.environ.get("STCAT_NO_FFN_PLANES")
FFN_PLANES_FULL = not os.environ.get("STCAT_NO_FFN_PLANES_FULL")
QK_PLANES = not os.environ.get("STCAT_NO_QK_PLANES")    # (the spatial layers' q / k in-projection on the A-stationary kernel)    # (linear2 and both weight gradients on the plane kernels too)


class FfnPlanes:
    """per-encoder cache for the FFN's wide side on the plane kernels (round 5): weight planes of linear1 (forward form)
    and linear2 (transposed form) of the layers whose token matrix is big enough, refreshed by ONE launch per forward pass,
    plus the constant 1 / (1 - p) vectors of the masked data gradient"""

    def __init__(self):
        self.wp = ops.WeightPlanes()
        self.gain = {}

    def refresh(self, pairs, need_bwd: bool, extra=()):
        """pairs: [(W1 [F, D], W2 [D, F]), ...] -> {W1.data_ptr(): (planes of W1, of W1^T, of W2, of W2^T)};
        extra: further [N, K] weights (the q / k rows of the in-projections) -> their forward planes under their data_ptr"""
        ws = []
        for W1, W2 in pairs:
            ws.append(W1.view(W1.shape[0], 1, 1, W1.shape[1]))
            ws.append(W2.view(W2.shape[0], 1, 1, W2.shape[1]))
        for W in extra:
            ws.append(W.view(W.shape[0], 1, 1, W.shape[1]))
        fwd, tr = self.wp.refresh(ws, transposed=need_bwd)
        out = {W1.data_ptr(): (fwd[W1.data_ptr()], tr.get(W1.data_ptr()) if need_bwd else None, fwd[W2.data_ptr()],
                               tr.get(W2.data_ptr()) if need_bwd else None) for W1, W2 in pairs}
        for W in extra:
            out[("qk", W.data_ptr())] = fwd[W.data_ptr()]
        return out

    def gainvec(self, like, n: int, gain: float):
        key = (str(like.device), n, round(gain, 9))
        v = self.gain.get(key)
        if v is None:
            if L.RECORDER is not None:
                return None            # (cannot be created inside a recording: the eager first call makes it)
            v = self.gain[key] = torch.full((n,), gain, device=like.device, dtype=torch.float32)
        return v


def _ffn_f(x, W1, b1, W2, b2, g, be, p, pl=None):
    """norm(x + dropout(linear2(dropout(relu(linear1 x)))))  (modal_encoder.py:239-241; query_decoder.py:435-437, 657-659).
    Round 5: the ReLU AND the inner dropout ride in linear1's epilogue (stcat_linear_fwd_drop), their backward in the
    epilogue of linear2's data gradient (stcat_linear_dgrad_mask): per layer one [M, 2048] pass less forward (the dropout
    launch) and two less backward (dropout + ReLU backward) — 125 us per spatial encoder layer at C3 — and the pre-dropout
    activation is no longer kept.  The split-bf16 modes only; mma mode f32 keeps the separate launches."""
    if FUSE_FFN and L.get_mma_mode() != "f32" and W1.shape[0] % 64 == 0 and W2.shape[0] % 64 == 0:
        shp = x.shape
        K = shp[-1]
        x2 = x if x.dim() == 2 else x.reshape(-1, K)
        if not (x2.is_contiguous() or (x2.stride(1) == 1 and x2.stride(0) % 4 == 0)):
            x2 = x2.contiguous()
        M, N = x2.shape[0], W1.shape[0]
        ent = pl[0].get(W1.data_ptr()) if pl is not None else None
        gv = pl[1].gainvec(x, N, 1.0 / (1.0 - p) if p > 0.0 else 1.0) if ent is not None else None
        if ent is not None and gv is not None and x2.is_contiguous():
            # linear1 on the A-stationary plane kernel (K = 256 -> N = 2048: its shape), operands as planes
            w1p, w1t, w2p, w2t = ent
            xp = ops.pl_split(x2)
            ymask = torch.empty(M, N // 8, device=x.device, dtype=torch.uint8)
            seed, off, base = ops._dropout_stream.take(M * N, x.device) if p > 0.0 else (0, 0, None)
            st_x = L.stream_of(x2)
            if FFN_PLANES_FULL and W2.shape[0] % 64 == 0:
                # ... its result stays in planes: linear2 (K = 2048 -> 256) runs on the plane tile kernel, and the backward
                # pass reads the planes again (linear2's weight gradient); the fp32 [M, 2048] activation is never written
                hp = ops.Planes.empty(x, M, N)
                L.call("stcat_pl_linear_fwd", xp.h, xp.l, w1p.h, w1p.l, L._ptr(b1), None, None, hp.h, hp.l, ymask.data_ptr(),
                       M, N, K, 1, float(p), seed, off, base, st_x)
                D = W2.shape[0]
                res2 = x.reshape(M, D)
                res2 = res2 if res2.is_contiguous() else res2.contiguous()
                h2 = torch.empty(M, D, device=x.device, dtype=torch.float32)
                L.call("stcat_pl_linear_fwd", hp.h, hp.l, w2p.h, w2p.l, L._ptr(b2), res2.data_ptr() if p == 0.0 else None,
                       h2.data_ptr(), None, None, None, M, D, N, 0, 0.0, 0, 0, None, st_x)
                y, c_n = _f(ops.LayerNormFn, _T, h2.view(*shp[:-1], D), x if p > 0.0 else None, g, be, 1e-5, p)
                return y, ("planes", c_n, xp, hp, ymask, gv, (w1t, w2t), (W1, W2), p, y.shape)
            f1d = torch.empty(M, N, device=x.device, dtype=torch.float32)
            L.call("stcat_pl_linear_fwd", xp.h, xp.l, w1p.h, w1p.l, L._ptr(b1), None, f1d.data_ptr(), None, None,
                   ymask.data_ptr(), M, N, K, 1, float(p), seed, off, base, st_x)
            y, st = _outln_f(f1d.view(*shp[:-1], N), W2, b2, x, g, be, p)
            return y, (st, ("bits", ymask, gv, w2t), x2, f1d, W1)
        if p > 0.0:
            seed, off, base = ops._dropout_stream.take(M * N, x.device)
            f1d = torch.empty(M, N, device=x.device, dtype=torch.float32)
            L.call("stcat_linear_fwd_drop", x2.data_ptr(), W1.data_ptr(), L._ptr(b1), None, f1d.data_ptr(), M, N, K,
                   x2.stride(0), N, 0, 1, float(p), seed, off, base, L.stream_of(x2))
        else:
            f1d = ops.linear_fwd_raw(x2, W1, b1, None, True)
        y, st = _outln_f(f1d.view(*shp[:-1], N), W2, b2, x, g, be, p)
        return y, (st, ("fused", 1.0 / (1.0 - p) if p > 0.0 else 1.0), x2, f1d, W1)
    f1, x_1 = _lin_f(x, W1, b1, relu=True)
    c_dr = None
    f1d = f1
    if p > 0.0:
        f1d, c_dr = _f(ops.DropoutFn, _T, f1, None, p)
    y, st = _outln_f(f1d, W2, b2, x, g, be, p)
    return y, (st, c_dr, x_1, f1, W1)


def _ffn_b(st, dy):
    """-> (d_x [M,D], dW1, db1, dW2, db2, dg, dbe)"""
    if st[0] == "planes":
        return _ffn_b_planes(st, dy)
    st_o, c_dr, x_1, f1, W1 = st
    if isinstance(c_dr, tuple) and c_dr[0] == "bits":
        d_f1, d_x_res, dW2, db2, dg, dbe = _outln_b(st_o, dy, mask=c_dr)
        d_x, dW1, db1, _ = _lin_b(d_f1, x_1, W1, add=d_x_res)
        return d_x, dW1, db1, dW2, db2, dg, dbe
    if isinstance(c_dr, tuple) and c_dr[0] == "fused":
        d_f1, d_x_res, dW2, db2, dg, dbe = _outln_b(st_o, dy, mask=(f1, c_dr[1]))     # f1 = dropout(relu(.)) here
        d_x, dW1, db1, _ = _lin_b(d_f1, x_1, W1, add=d_x_res)
        return d_x, dW1, db1, dW2, db2, dg, dbe
    d_f1d, d_x_res, dW2, db2, dg, dbe = _outln_b(st_o, dy)
    d_f1 = ops.DropoutFn.backward(c_dr, d_f1d.view(f1.shape))[0] if c_dr is not None else d_f1d
    d_x, dW1, db1, _ = _lin_b(d_f1, x_1, W1, relu_y=f1, add=d_x_res)      # + residual gradient, fused into the dgrad
    return d_x, dW1, db1, dW2, db2, dg, dbe


def _ffn_b_planes(st, dy):
    """the FFN's backward with every [M, 2048] tensor in planes (round 5): LayerNorm backward (fp32) -> split -> linear2's data
    gradient with the ReLU + dropout backward from the bit mask (A-stationary kernel, planes out) -> linear1's data gradient
    (plane tile kernel, K = 2048, fp32 out + the residual gradient); both weight gradients on the plane weight-gradient kernel,
    the bias gradients as column sums — all four on the weight-gradient stream."""
    _, c_n, xp, hp, ymask, gv, (w1t, w2t), (W1, W2), p, shp = st
    r = ops.LayerNormFn.backward(c_n, dy.reshape(shp))
    d_h, d_res, dg, dbe = r[0], r[1], r[2], r[3]
    if p == 0.0:
        d_res = d_h
    F, D = W1.shape
    d_h2 = d_h.reshape(-1, D)
    d_h2 = d_h2 if d_h2.is_contiguous() else d_h2.contiguous()
    M = d_h2.shape[0]
    stx = L.stream_of(d_h2)
    gp = ops.pl_split(d_h2)
    d_f1 = ops.Planes.empty(d_h2, M, F)
    L.call("stcat_pl_linear_dgrad_mask", gp.h, gp.l, w2t.h, w2t.l, ymask.data_ptr(), gv.data_ptr(), None, d_f1.h, d_f1.l,
           M, D, F, stx)
    add = d_res.reshape(M, D)
    add = add if add.is_contiguous() else add.contiguous()
    d_x = torch.empty(M, D, device=d_h2.device, dtype=torch.float32)
    L.call("stcat_pl_linear_fwd", d_f1.h, d_f1.l, w1t.h, w1t.l, None, add.data_ptr(), d_x.data_ptr(), None, None, None,
           M, D, F, 0, 0.0, 0, 0, None, stx)
    dW1, db1 = ops._zeros(d_h2, F, D), ops._zeros(d_h2, F)
    dW2, db2 = ops._zeros(d_h2, D, F), ops._zeros(d_h2, D)

    def view4(pl, C):
        return ops.Planes(pl.t.view(pl.t.shape[0], 1, 1, M, C))

    def wgrads():
        ops.pl_conv_wgrad_raw(view4(gp, D), view4(hp, F), (D, 1, 1, F), 1, 0, out=dW2.view(D, 1, 1, F))
        ops.pl_conv_wgrad_raw(view4(d_f1, F), view4(xp, D), (F, 1, 1, D), 1, 0, out=dW1.view(F, 1, 1, D))
        L.call("stcat_colsum", d_h2.data_ptr(), None, db2.data_ptr(), M, D, L.stream_of(d_h2))
        L.call("stcat_pl_colsum", d_f1.h, d_f1.l, db1.data_ptr(), M, F, L.stream_of(d_h2))

    keep = (gp.t, hp.t, d_f1.t, xp.t, d_h2)
    if _BATCH and _BATCH[-1] is not None and DEFER_WGRADS:
        _BATCH[-1].append((("call", wgrads, keep), gp.t, hp.t))
    else:
        wgrads()
    return d_x, dW1, db1, dW2, db2, dg, dbe


def _ln_f(x, g, be, out=None):
    M, D = x.shape
    y = out if out is not None else torch.empty_like(x)
    mean = ops._empty(x, M)
    rstd = ops._empty(x, M)
    L.call("stcat_layernorm_fwd", x.data_ptr(), None, g.data_ptr(), be.data_ptr(), y.data_ptr(), mean.data_ptr(),
           rstd.data_ptr(), M, D, 1e-5, 0.0, 0, 0, None, L.stream_of(x))
    return y, (x, g, mean, rstd)


def _ln_b(st, dy, dg, dbe):
    """dz; dg / dbe accumulate (the kernel adds into caller-zeroed buffers)"""
    x, g, mean, rstd = st
    M, D = x.shape
    dy = dy if dy.is_contiguous() else dy.contiguous()
    dz = torch.empty_like(x)
    L.call("stcat_layernorm_bwd", dy.data_ptr(), x.data_ptr(), None, g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
           dz.data_ptr(), None, dg.data_ptr(), dbe.data_ptr(), M, D, 0.0, 0, 0, None, L.stream_of(x))
    return dz


# ------------------------------------------------------------------------------------------------------------------
# encoder layer (modal_encoder.py:207-242), post-norm: 12 of them per step
# ------------------------------------------------------------------------------------------------------------------
class EncoderLayerFn(Function):
    @staticmethod
    def forward(ctx, x, pos, kpm, p, nhead, W_in, B_in, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2, ffn_pl=None):
        D = x.shape[-1]
        shp = x.shape
        x = x if x.is_contiguous() else x.contiguous()
        pos_b = pos if pos.shape == x.shape else pos.expand_as(x)
        wqk = ffn_pl[0].get(("qk", W_in.data_ptr())) if (ffn_pl is not None and QK_PLANES) else None
        if wqk is not None:
            # q = k = src + pos as planes in ONE pass, the [M, 2D] in-projection on the A-stationary plane kernel (round 5)
            qk_in, qkp = ops.pl_split_sum(x, pos_b if pos_b.is_contiguous() else pos_b.contiguous())
            M_ = qk_in.numel() // D
            qk = torch.empty(*shp[:-1], 2 * D, device=x.device, dtype=torch.float32)
            L.call("stcat_pl_linear_fwd", qkp.h, qkp.l, wqk.h, wqk.l, B_in[:2 * D].data_ptr(), None, qk.data_ptr(), None, None,
                   None, M_, 2 * D, D, 0, 0.0, 0, 0, None, L.stream_of(x))
            x_qk = qk_in.view(M_, D)
        else:
            qk_in = ops.ew(L.EW_ADD, x, pos_b if pos_b.is_contiguous() else pos_b.contiguous())      # q = k = src + pos :234
            qk, x_qk = _lin_f(qk_in, W_in[:2 * D], B_in[:2 * D])
        v, x_v = _lin_f(x, W_in[2 * D:], B_in[2 * D:])
        (a, _), c_att = _f(ops.MhaSelfFn, (True, False, True) + (False,) * 5, qk, qk[:, :, D:], v, kpm,
                           (D // nhead) ** -0.5, False, True, p)
        x1, st1 = _outln_f(a, Wo, bo, x, g1, be1, p)
        y, st2 = _ffn_f(x1, W1, b1, W2, b2, g2, be2, p, pl=ffn_pl)
        ctx.st = (c_att, st1, st2, x_qk, x_v, D, shp, pos.shape)
        ctx.W_in = W_in
        return y

    @staticmethod
    def backward(ctx, dy):
        with wgrad_batch(dy):
            return EncoderLayerFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        (c_att, st1, st2, x_qk, x_v, D, shp, pos_shape) = ctx.st
        W_in = ctx.W_in
        need_x, need_pos = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_x1, dW1, db1, dW2, db2, dg2, dbe2 = _ffn_b(st2, dy)
        d_a, d_x_res, dWo, dbo, dg1, dbe1 = _outln_b(st1, d_x1)
        r = ops.MhaSelfFn.backward(c_att, d_a.view(shp), None)
        dqk, dv = r[0], r[2]
        dW_in = ops._zeros(dy, 3 * D, D)        # packed in-projection gradient: the two GEMMs write its row blocks
        dB_in = ops._zeros(dy, 3 * D)
        d_x, _, _, _ = _lin_b(dv, x_v, W_in[2 * D:], need_dx=need_x, dw=dW_in[2 * D:], db=dB_in[2 * D:],
                              add=d_x_res if need_x else None)
        d_pos = None
        if need_pos:
            d_qkin, _, _, _ = _lin_b(dqk, x_qk, W_in[:2 * D], dw=dW_in[:2 * D], db=dB_in[:2 * D])
            if need_x:
                d_x = _add(d_x, d_qkin)
            d_pos = d_qkin.view(shp)
            if tuple(pos_shape) != tuple(shp):
                d_pos = d_pos.sum_to_size(pos_shape)
        else:
            d_x, _, _, _ = _lin_b(dqk, x_qk, W_in[:2 * D], need_dx=need_x, dw=dW_in[:2 * D], db=dB_in[:2 * D],
                                  add=d_x if need_x else None)
        return ((d_x.view(shp) if d_x is not None else None), d_pos, None, None, None, dW_in, dB_in, dWo, dbo, dg1, dbe1,
                dW1, db1, dW2, db2, dg2, dbe2)


def encoder_layer(layer, x, pos, kpm, pos_is_const: bool):
    a = layer.self_attn
    p = layer.dropout_p if layer.training else 0.0
    if pos_is_const:
        pos = pos.detach()
    return EncoderLayerFn.apply(x, pos, kpm, p, layer.nhead, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight,
                                a.out_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.weight,
                                layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm2.weight,
                                layer.norm2.bias)


# ------------------------------------------------------------------------------------------------------------------
# time decoder: all layers (query_decoder.py:478-550, 553-660) as ONE node
# ------------------------------------------------------------------------------------------------------------------
_NT_LAYER = 18


class TimeDecoderFn(Function):
    """inputs: memory [n,S',D], pos [n,S',D] (constant), kpm, query_pos [T,D], time_pos [T,D] (constant) ->
    (hs [L,T,D], head-mean self-attention weights [L,1,T,T]).  Each layer projects (memory + pos) / memory with the
    key / value rows of its packed cross_attn_image in-projection (:633-639); the packed gradients are written in place."""

    @staticmethod
    def forward(ctx, memory, pos, kpm, query_pos, time_pos, p, nhead, nl, gN, beN, *prm):
        T, D = query_pos.shape
        hd = D // nhead
        memory = memory if memory.is_contiguous() else memory.contiguous()
        if not query_pos.is_contiguous():     # the template's content row, expanded over the frames (:463): our own copy
            query_pos = ops.ew2d(L.EW_COPY, query_pos) if query_pos.stride(1) == 1 else query_pos.contiguous()
        mem_pos = ops.ew(L.EW_ADD, memory, pos)                                               # :636
        qpos_time = ops.ew(L.EW_ADD, query_pos, time_pos)                                     # :602
        hs = ops._empty(memory, nl, T, D)
        ws = ops._empty(memory, nl, 1, T, T)
        out = ops._zeros(memory, T, D)
        states = []
        x_mp = x_mem = None
        for i in range(nl):
            (W_in, B_in, Wo, bo, g1, be1, W_c, B_c, Wo2, bo2, g3, be3, W1, b1, W2, b2, g4, be4) = \
                prm[i * _NT_LAYER:(i + 1) * _NT_LAYER]
            st = {}
            qk_in = ops.ew(L.EW_ADD, out, qpos_time)
            qk, st["x_qk"] = _lin_f(qk_in, W_in[:2 * D], B_in[:2 * D])
            v, st["x_v"] = _lin_f(out, W_in[2 * D:], B_in[2 * D:])
            (a, w), st["att"] = _f(ops.MhaSelfFn, (True, False, True) + (False,) * 5, qk[None], qk[None][:, :, D:], v[None],
                                   None, hd ** -0.5, True, True, p)                          # :604-610
            ops.ew(L.EW_COPY, w, out=ws[i])
            tgt1, st["o1"] = _outln_f(a[0], Wo, bo, out, g1, be1, p)
            qc_in = ops.ew(L.EW_ADD, tgt1, query_pos)                                         # :633-634
            qc, st["x_qc"] = _lin_f(qc_in, W_c[:D], B_c[:D])
            kci, x_mp = _lin_f(mem_pos, W_c[D:2 * D], B_c[D:2 * D])
            vvi, x_mem = _lin_f(memory, W_c[2 * D:], B_c[2 * D:])
            a2, st["q1"] = _f(ops.AttnQ1Fn, _T, qc, None, kci, None, vvi, kpm, hd ** -0.5, p)
            tgt2, st["o3"] = _outln_f(a2, Wo2, bo2, tgt1, g3, be3, p)                         # :653-654
            out, st["ffn"] = _ffn_f(tgt2, W1, b1, W2, b2, g4, be4, p)                         # :657-659
            _, st["norm"] = _ln_f(out, gN, beN, out=hs[i])
            states.append(st)
        ctx.states = states
        ctx.prm = prm
        ctx.gN = gN
        ctx.mem = (x_mem, x_mp, memory.shape)
        ctx.dims = (T, D, nl)
        return hs, ws

    @staticmethod
    def backward(ctx, d_hs, d_ws):
        # Round 5: this node's weight gradients stay ON its own chain (no hand-over to the shared weight-gradient stream).
        # The node runs on the forked stream beside the box decoder, whose ~330 weight-gradient launches fill that stream
        # first (the host replays the box decoder's backward before this one): this node's ~100 launches queued behind them
        # and its final join — which the encoder's backward waits for — came 0.9 ms after the box decoder had finished
        # (tools/node_times.py: 3.75 ms against 2.86 ms).  Inline they lengthen this chain to ~2.8 ms, still the shorter one.
        if INLINE_TIME_WGRADS:
            _BATCH.append(None)          # (a None frame: _wgrad launches directly, nested flushes are no-ops)
            try:
                return TimeDecoderFn._backward(ctx, d_hs, d_ws)
            finally:
                _BATCH.pop()
        with wgrad_batch(d_hs):
            return TimeDecoderFn._backward(ctx, d_hs, d_ws)

    @staticmethod
    def _backward(ctx, d_hs, d_ws):
        T, D, nl = ctx.dims
        prm = ctx.prm
        x_mem, x_mp, mshape = ctx.mem
        need_mem, need_qpos = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        like = d_hs
        d_hs = d_hs if d_hs.is_contiguous() else d_hs.contiguous()
        dgN = ops._zeros(like, D)
        dbeN = ops._zeros(like, D)
        d_mem = d_mp = d_qpos = d_qpt = d_next = None
        d_layers = [None] * nl
        for i in reversed(range(nl)):
            st = ctx.states[i]
            (W_in, B_in, Wo, bo, g1, be1, W_c, B_c, Wo2, bo2, g3, be3, W1, b1, W2, b2, g4, be4) = \
                prm[i * _NT_LAYER:(i + 1) * _NT_LAYER]
            d_out = _ln_b(st["norm"], d_hs[i], dgN, dbeN)
            if d_next is not None:
                d_out = _add(d_out, d_next)
            d_tgt2, dW1, db1, dW2, db2, dg4, dbe4 = _ffn_b(st["ffn"], d_out)
            d_a2, d_tgt1_res, dWo2, dbo2, dg3, dbe3 = _outln_b(st["o3"], d_tgt2)
            r = ops.AttnQ1Fn.backward(st["q1"], d_a2)
            dqc, dkc, dvv = r[0], r[2], r[4]
            dW_c = ops._zeros(like, 3 * D, D)
            dB_c = ops._zeros(like, 3 * D)
            d_mp, _, _, _ = _lin_b(dkc, x_mp, W_c[D:2 * D], need_dx=need_mem, dw=dW_c[D:2 * D], db=dB_c[D:2 * D], add=d_mp)
            d_mem, _, _, _ = _lin_b(dvv, x_mem, W_c[2 * D:], need_dx=need_mem, dw=dW_c[2 * D:], db=dB_c[2 * D:], add=d_mem)
            d_qcin, _, _, _ = _lin_b(dqc, st["x_qc"], W_c[:D], dw=dW_c[:D], db=dB_c[:D])     # gradient of (tgt1 + query_pos)
            d_qpos = d_qcin if d_qpos is None else _add(d_qpos, d_qcin)
            d_tgt1 = _add(d_tgt1_res.reshape(T, D), d_qcin)
            d_a, d_out_res, dWo, dbo, dg1, dbe1 = _outln_b(st["o1"], d_tgt1)
            r = ops.MhaSelfFn.backward(st["att"], d_a.view(1, T, D), d_ws[i] if d_ws is not None else None)
            dqk, dv = r[0], r[2]
            dW_in = ops._zeros(like, 3 * D, D)
            dB_in = ops._zeros(like, 3 * D)
            first = i == 0                               # layer 0's input state is the constant zero tensor
            d_t, _, _, _ = _lin_b(dv[0], st["x_v"], W_in[2 * D:], need_dx=not first, dw=dW_in[2 * D:], db=dB_in[2 * D:],
                                  add=None if first else d_out_res.reshape(T, D))
            d_qkin, _, _, _ = _lin_b(dqk[0], st["x_qk"], W_in[:2 * D], dw=dW_in[:2 * D], db=dB_in[:2 * D])
            d_qpt = d_qkin if d_qpt is None else _add(d_qpt, d_qkin)
            d_next = None if first else _add(d_t, d_qkin)
            d_layers[i] = (dW_in, dB_in, dWo, dbo, dg1, dbe1, dW_c, dB_c, dWo2, dbo2, dg3, dbe3, dW1, db1, dW2, db2, dg4, dbe4)
            wgrad_flush(like)
        if need_mem:
            d_mem = _add(d_mem, d_mp).view(mshape)
        d_qp = _add(d_qpos, d_qpt) if need_qpos else None
        flat = ()
        for t in d_layers:
            flat += t
        return (d_mem if need_mem else None, None, None, d_qp, None, None, None, None, dgN, dbeN) + flat


def time_decoder(dec, memory, pos, kpm, query_pos, time_pos):
    """dec: grounding.TimeDecoder"""
    p = dec.layers[0].dropout_p if dec.training else 0.0
    wb = lambda m: (m.weight, m.bias)       # noqa: E731
    prm = ()
    for l in dec.layers:
        sa, ca = l.self_attn, l.cross_attn_image
        prm += ((sa.in_proj_weight, sa.in_proj_bias) + wb(sa.out_proj) + wb(l.norm1) + (ca.in_proj_weight, ca.in_proj_bias)
                + wb(ca.out_proj) + wb(l.norm3) + wb(l.linear1) + wb(l.linear2) + wb(l.norm4))
    return plans.apply(TimeDecoderFn, memory, pos, kpm, query_pos, time_pos, p, dec.layers[0].nhead, len(dec.layers),
                               dec.norm.weight, dec.norm.bias, *prm)


# ------------------------------------------------------------------------------------------------------------------
# box decoder: all layers + the anchor-update loop around them (query_decoder.py:150-247, 250-438) as ONE node
# ------------------------------------------------------------------------------------------------------------------
_N_SHARED = 16      # ref_point_head (W,b)x2, query_scale (W,b)x2, bbox_embed (W,b)x3, norm (g,b)
_N_LAYER = 42


class BoxDecoderFn(Function):
    """inputs: memory / pos [n,S',D] (pos: the constant sine embedding), kpm, anchor [T,4], time_embed [T,D] ->
    (hs [L,T,D], refs [L,T,4]).  Each layer projects the memory itself (ca_kcontent_proj / ca_kpos_proj / ca_v_proj,
    query_decoder.py:355-358), so the key / value gradients feed that layer's weight and data gradients directly.
    Shared modules (ref_point_head, query_scale, bbox_embed, norm) accumulate their gradients over the layers inside
    the node instead of through L AccumulateGrad adds."""

    @staticmethod
    def forward(ctx, memory, pos, kpm, anchor, time_embed, p, nhead, nl, *prm):
        T = anchor.shape[0]
        memory = memory if memory.is_contiguous() else memory.contiguous()
        pos = pos if pos.is_contiguous() else pos.contiguous()
        D = time_embed.shape[1]
        hd = D // nhead
        sh = prm[:_N_SHARED]
        (Wr1, br1, Wr2, br2, Ws1, bs1, Ws2, bs2, Wb1, bb1, Wb2, bb2, Wb3, bb3, gN, beN) = sh
        anchor = anchor.contiguous()
        time_embed = time_embed.contiguous()
        hs = ops._empty(anchor, nl, T, D)
        refs = ops._empty(anchor, nl, T, 4)
        ops.ew(L.EW_COPY, anchor, out=refs[0])
        out = ops._zeros(anchor, T, D)
        states = []
        # Round 5: the memory-side projections of a layer (ca_kpos_proj(pos), ca_v_proj(memory), ca_kcontent_proj(memory),
        # query_decoder.py:355-358 — three [n*S', 256] x 256 GEMMs, 72 us of a layer's 370 us chain at C3) do not depend on
        # the query state: layer i + 1's run on a second lane (the weight-gradient stream, idle in the forward pass) while
        # the chain works through layer i; the chain waits for them right before its cross-attention.
        lane = _Lane(memory, 1, PROJ_LANE)
        x_pos, x_mem = pos.reshape(-1, D), memory.reshape(-1, D)
        rows = x_mem.shape[0]

        def mem_proj(i):
            lpm = prm[_N_SHARED + i * _N_LAYER + 36: _N_SHARED + (i + 1) * _N_LAYER]
            (Wmk_, bmk_, Wmp_, bmp_, Wmv_, bmv_) = lpm
            kpi_, vvi_, kci_ = (ops._empty(memory, rows, D) for _ in range(3))     # (allocated on the chain's stream)
            # The lane is ordered behind the chain HERE, after the allocations: the caching allocator may
            # hand back a block whose last kernel — a temporary of layer i freed on the host — is still queued on the
            # chain's stream; an idle lane would write it first and the late chain kernel would overwrite the projection.
            # The chain has just waited for the lane (sync_main), so this costs no overlap.
            lane.fork()
            with lane:
                ops.linear_fwd_raw(x_pos, Wmp_, bmp_, out=kpi_)                                # :355-358
                ops.linear_fwd_raw(x_mem, Wmv_, bmv_, out=vvi_)
                ops.linear_fwd_raw(x_mem, Wmk_, bmk_, kpi_ if i == 0 else None, out=kci_)      # first layer: + k_pos :360-366
            lane.keep(kpi_, vvi_, kci_)
            return (kci_.view(memory.shape), kpi_.view(memory.shape), vvi_.view(memory.shape))

        nxt = mem_proj(0)
        for i in range(nl):
            lp = prm[_N_SHARED + i * _N_LAYER: _N_SHARED + (i + 1) * _N_LAYER]
            (Wqc, bqc, Wqp, bqp, Wqt, bqt, Wkc, bkc, Wkp, bkp, Wkt, bkt, Wv, bv, W_in, B_in, Wo, bo, g1, be1, Wcq, bcq,
             Wcqp, bcqp, Wqs, bqs, Wo2, bo2, g3, be3, W1, b1, W2, b2, g4, be4, Wmk, bmk, Wmp, bmp, Wmv, bmv) = lp
            first = i == 0
            st = {}
            sine, st["sine"] = _f(ops.SineEmbedFn, _T, anchor)                                # [T,512]  :190
            h_r, st["x_r1"] = _lin_f(sine, Wr1, br1, relu=True)                               # ref_point_head :191
            qpos, st["x_r2"] = _lin_f(h_r, Wr2, br2)
            st["h_r"] = h_r
            sine_h = sine[:, :D]
            if first:
                sine_q = sine_h                                                              # row-strided view
            else:                                                                            # :194-200
                h_s, st["x_s1"] = _lin_f(out, Ws1, bs1, relu=True)
                qsc, st["x_s2"] = _lin_f(h_s, Ws2, bs2)
                st["h_s"] = h_s
                sine_q = ops.ew2d(L.EW_MUL, sine_h, qsc)
            st["sine_h"] = sine_h
            # ---- the layer: self-attention over the T queries :329-345
            mg = ops.linear_multi_ok(T, D, D, out)     # grouped launches for the independent skinny projections (round 5)
            if mg:
                # q = Wqc out + Wqt time + Wqp pos, k likewise, v = Wv out: seven problems, three (zeroed) outputs, ONE launch
                q, k, v = ops._zeros(out, T, D), ops._zeros(out, T, D), ops._zeros(out, T, D)
                ops.linear_fwd_multi([out, time_embed, qpos, out, time_embed, qpos, out], [Wqc, Wqt, Wqp, Wkc, Wkt, Wkp, Wv],
                                     [bqc, bqt, bqp, bkc, bkt, bkp, bv], [q, q, q, k, k, k, v], T, D, D)
                qp, kp_, vp = ops._zeros(out, T, D), ops._zeros(out, T, D), ops._zeros(out, T, D)
                ops.linear_fwd_multi([q, k, v], [W_in[:D], W_in[D:2 * D], W_in[2 * D:]],
                                     [B_in[:D], B_in[D:2 * D], B_in[2 * D:]], [qp, kp_, vp], T, D, D)
            else:
                t, _ = _lin_f(out, Wqc, bqc)
                t, _ = _lin_f(time_embed, Wqt, bqt, res=t)
                q, _ = _lin_f(qpos, Wqp, bqp, res=t)
                t, _ = _lin_f(out, Wkc, bkc)
                t, _ = _lin_f(time_embed, Wkt, bkt, res=t)
                k, _ = _lin_f(qpos, Wkp, bkp, res=t)
                v, _ = _lin_f(out, Wv, bv)
                qp, _ = _lin_f(q, W_in[:D], B_in[:D])
                kp_, _ = _lin_f(k, W_in[D:2 * D], B_in[D:2 * D])
                vp, _ = _lin_f(v, W_in[2 * D:], B_in[2 * D:])
            (a, _), st["att"] = _f(ops.MhaSelfFn, _T, qp[None], kp_[None], vp[None], None, hd ** -0.5, False, False, p)
            tgt1, st["o1"] = _outln_f(a[0], Wo, bo, out, g1, be1, p)
            st["sa_in"] = (out, qpos, q, k, v)
            # ---- time-aligned cross-attention :355-432
            if mg and sine_q.is_contiguous():
                qc, qs = ops._zeros(out, T, D), ops._zeros(out, T, D)
                xs_, ws_, bs_, ys_ = [tgt1, sine_q], [Wcq, Wqs], [bcq, bqs], [qc, qs]
                if first:                                                                    # :360-366
                    xs_.append(qpos); ws_.append(Wcqp); bs_.append(bcqp); ys_.append(qc)
                ops.linear_fwd_multi(xs_, ws_, bs_, ys_, T, D, D)
                st["x_qs"] = sine_q
            else:
                qc, _ = _lin_f(tgt1, Wcq, bcq)
                if first:                                                                    # :360-366
                    qc, _ = _lin_f(qpos, Wcqp, bcqp, res=qc)
                qs, st["x_qs"] = _lin_f(sine_q, Wqs, bqs)                                    # :369
            kci, kpi, vvi = nxt
            lane.sync_main()                       # this layer's memory-side projections (queued a layer ago) are in
            if i + 1 < nl:
                nxt = mem_proj(i + 1)              # ... the next layer's start now, beside the rest of this layer
            a2, st["q1"] = _f(ops.AttnQ1Fn, _T, qc, qs, kci, kpi, vvi, kpm, (2 * hd) ** -0.5, p)
            tgt2, st["o3"] = _outln_f(a2, Wo2, bo2, tgt1, g3, be3, p)
            st["tgt1"] = tgt1
            out, st["ffn"] = _ffn_f(tgt2, W1, b1, W2, b2, g4, be4, p)                          # :435-437
            # ---- anchor update :212-219 (the last layer's update is never read: not computed)
            if i != nl - 1:
                e1, st["x_b1"] = _lin_f(out, Wb1, bb1, relu=True)
                e2, st["x_b2"] = _lin_f(e1, Wb2, bb2, relu=True)
                tmp, st["x_b3"] = _lin_f(e2, Wb3, bb3)
                st["e"] = (e1, e2)
                pre = ops.ew(L.EW_ADD, tmp, ops.ew(L.EW_INVSIG, anchor))
                ops.ew(L.EW_SIGMOID, pre, out=refs[i + 1])
                st["anchor"] = anchor
                anchor = refs[i + 1]
            _, st["norm"] = _ln_f(out, gN, beN, out=hs[i])                                   # :221-229
            states.append(st)
        ctx.states = states
        ctx.prm = prm
        ctx.time_embed = time_embed
        ctx.mem = (x_mem, x_pos, memory.shape)
        ctx.dims = (T, D, nl)
        ctx.refs = refs
        # ---- box head on the normalised states of all layers (pipeline.py:88-93): same bbox_embed as the anchor update
        # (a DETACHED alias of hs: a view of the output itself, kept in ctx.head, would tie this node to its own output through
        #  the view's base — a reference cycle that only backward() opens; a forward that is never back-propagated then keeps
        #  the whole graph, the backbone's tape included, alive for the life of the process: round-4 memory log)
        e1, x_h1 = _lin_f(hs.detach().view(nl * T, D), Wb1, bb1, relu=True)
        e2, x_h2 = _lin_f(e1, Wb2, bb2, relu=True)
        tmp, x_h3 = _lin_f(e2, Wb3, bb3)
        coord = ops.ew(L.EW_SIGMOID, ops.ew(L.EW_ADD, tmp, ops.ew(L.EW_INVSIG, refs.view(nl * T, 4))))
        ctx.head = (x_h1, x_h2, x_h3, e1, e2, coord)
        # (copies: the node keeps `refs` / `coord`, and an output held by its own node is a reference cycle)
        return hs, ops.ew(L.EW_COPY, refs), ops.ew(L.EW_COPY, coord).view(nl, T, 4)

    @staticmethod
    def backward(ctx, d_hs, d_refs, d_coord):
        with wgrad_batch(d_hs):
            return BoxDecoderFn._backward(ctx, d_hs, d_refs, d_coord)

    @staticmethod
    def _backward(ctx, d_hs, d_refs, d_coord):
        T, D, nl = ctx.dims
        prm, refs = ctx.prm, ctx.refs
        x_mem, x_pos, mshape = ctx.mem
        need_anchor = ctx.needs_input_grad[3]
        need_mem = ctx.needs_input_grad[0]
        d_mem = None
        (Wr1, br1, Wr2, br2, Ws1, bs1, Ws2, bs2, Wb1, bb1, Wb2, bb2, Wb3, bb3, gN, beN) = prm[:_N_SHARED]
        like = d_hs
        z = lambda t: ops._zeros(like, *t.shape)      # noqa: E731
        d_sh = [z(t) for t in prm[:_N_SHARED]]       # shared-module gradients: accumulated across the layers
        (dWr1, dbr1, dWr2, dbr2, dWs1, dbs1, dWs2, dbs2, dWb1, dbb1, dWb2, dbb2, dWb3, dbb3, dgN, dbeN) = d_sh
        d_hs = d_hs if d_hs.is_contiguous() else d_hs.contiguous()
        if d_refs is not None and not d_refs.is_contiguous():
            d_refs = d_refs.contiguous()
        if d_coord is not None:                          # box head
            x_h1, x_h2, x_h3, e1, e2, coord = ctx.head
            d_pre = ops.ew(L.EW_SIGMOID_BWD, d_coord.contiguous().view(nl * T, 4), coord)
            d_r = ops.ew(L.EW_INVSIG_BWD, d_pre, refs.view(nl * T, 4)).view(nl, T, 4)
            d_refs = d_r if d_refs is None else _add(d_refs, d_r)
            d_e2, _, _, _ = _lin_b(d_pre, x_h3, Wb3, dw=dWb3, db=dbb3)
            d_e1, _, _, _ = _lin_b(d_e2, x_h2, Wb2, dw=dWb2, db=dbb2, relu_y=e2)
            d_hs, _, _, _ = _lin_b(d_e1, x_h1, Wb1, dw=dWb1, db=dbb1, relu_y=e1, add=d_hs.view(nl * T, D))
            d_hs = d_hs.view(nl, T, D)
        d_layers = [None] * nl
        d_anchor = None
        d_next = None                                  # gradient reaching layer i's output state from layer i+1
        for i in reversed(range(nl)):
            st = ctx.states[i]
            lp = prm[_N_SHARED + i * _N_LAYER: _N_SHARED + (i + 1) * _N_LAYER]
            (Wqc, bqc, Wqp, bqp, Wqt, bqt, Wkc, bkc, Wkp, bkp, Wkt, bkt, Wv, bv, W_in, B_in, Wo, bo, g1, be1, Wcq, bcq,
             Wcqp, bcqp, Wqs, bqs, Wo2, bo2, g3, be3, W1, b1, W2, b2, g4, be4, Wmk, bmk, Wmp, bmp, Wmv, bmv) = lp
            first = i == 0
            d_out = _ln_b(st["norm"], d_hs[i], dgN, dbeN)
            if d_next is not None:
                d_out = _add(d_out, d_next)
            if i != nl - 1 and d_refs is not None:
                d_pre = ops.ew(L.EW_SIGMOID_BWD, d_refs[i + 1], refs[i + 1])
                if first and need_anchor:
                    d_anchor = ops.ew(L.EW_INVSIG_BWD, d_pre, st["anchor"])
                e1, e2 = st["e"]
                d_e2, _, _, _ = _lin_b(d_pre, st["x_b3"], Wb3, dw=dWb3, db=dbb3)
                d_e1, _, _, _ = _lin_b(d_e2, st["x_b2"], Wb2, dw=dWb2, db=dbb2, relu_y=e2)
                d_out, _, _, _ = _lin_b(d_e1, st["x_b1"], Wb1, dw=dWb1, db=dbb1, relu_y=e1, add=d_out)
            # ---- layer backward
            x_out, qpos, q, k, v = st["sa_in"]
            d_tgt2, dW1, db1, dW2, db2, dg4, dbe4 = _ffn_b(st["ffn"], d_out)
            d_a2, d_tgt1_res, dWo2, dbo2, dg3, dbe3 = _outln_b(st["o3"], d_tgt2)
            r = ops.AttnQ1Fn.backward(st["q1"], d_a2)
            dqc, dqs, dk1, dk2, dvv_i = r[0], r[1], r[2], r[3], r[4]
            d_mem, dWmk, dbmk, _ = _lin_b(dk1, x_mem, Wmk, need_dx=need_mem, add=d_mem)
            d_mem, dWmv, dbmv, _ = _lin_b(dvv_i, x_mem, Wmv, need_dx=need_mem, add=d_mem)
            _, dWmp, dbmp, _ = _lin_b(_add(dk1, dk2) if first else dk2, x_pos, Wmp, need_dx=False)
            mg = ops.linear_multi_ok(T, D, D, like) and st["x_qs"].is_contiguous()
            zTD = lambda: ops._zeros(like, T, D)        # noqa: E731
            zW = lambda: (ops._zeros(like, D, D), ops._zeros(like, D))     # noqa: E731
            nd = not first                              # layer 0's input state is the constant zero tensor
            d_qpos = None
            dWcqp = dbcqp = None
            if mg:
                # the independent data / weight gradients of the layer's skinny projections in grouped launches
                # (ops.linear_*_multi): cross-attention query side — three data gradients, three weight gradients
                dqc_, dqs_ = ops._c(dqc.reshape(T, D)), ops._c(dqs.reshape(T, D))
                d_tgt1 = zTD()
                gs_, ws_, ad_, dx_ = [dqc_], [Wcq], [ops._c(d_tgt1_res.reshape(T, D))], [d_tgt1]
                d_sine_q = None
                if (not first) or need_anchor:
                    d_sine_q = zTD()
                    gs_.append(dqs_); ws_.append(Wqs); ad_.append(None); dx_.append(d_sine_q)
                if first:
                    d_qpos = zTD()
                    gs_.append(dqc_); ws_.append(Wcqp); ad_.append(None); dx_.append(d_qpos)
                ops.linear_dgrad_multi(gs_, ws_, ad_, dx_, T, D, D)
                (dWcq, dbcq), (dWqs, dbqs) = zW(), zW()
                wg_ = [[dqc_, dqs_], [st["tgt1"], st["x_qs"]], [dWcq, dWqs], [dbcq, dbqs]]
                if first:
                    dWcqp, dbcqp = zW()
                    wg_[0].append(dqc_); wg_[1].append(qpos); wg_[2].append(dWcqp); wg_[3].append(dbcqp)
                _wgrad_multi(*wg_, T, D, D)
            else:
                d_sine_q, dWqs, dbqs, _ = _lin_b(dqs, st["x_qs"], Wqs, need_dx=(not first) or need_anchor)
                if first:
                    d_qpos, dWcqp, dbcqp, _ = _lin_b(dqc, qpos, Wcqp)
                d_tgt1, dWcq, dbcq, _ = _lin_b(dqc, st["tgt1"], Wcq, add=d_tgt1_res)
            d_a, d_out_res, dWo, dbo, dg1, dbe1 = _outln_b(st["o1"], d_tgt1)
            r = ops.MhaSelfFn.backward(st["att"], d_a.view(1, T, D), None)
            dW_in = ops._zeros(like, 3 * D, D)
            dB_in = ops._zeros(like, 3 * D)
            if mg:
                # ... the three in-projections of nn.MultiheadAttention ...
                gq, gk, gv = ops._c(r[0][0]), ops._c(r[1][0]), ops._c(r[2][0])
                d_q, d_k, d_v = zTD(), zTD(), zTD()
                ops.linear_dgrad_multi([gq, gk, gv], [W_in[:D], W_in[D:2 * D], W_in[2 * D:]], [None] * 3, [d_q, d_k, d_v], T, D, D)
                _wgrad_multi([gq, gk, gv], [q, k, v], [dW_in[:D], dW_in[D:2 * D], dW_in[2 * D:]],
                             [dB_in[:D], dB_in[D:2 * D], dB_in[2 * D:]], T, D, D)
                # ... and the seven input projections: d_qpos = d_q Wqp + d_k Wkp (+ the cross-attention's share in layer 0),
                # d_x = d_q Wqc + d_k Wkc + d_v Wv (+ the residual gradient)
                te = ctx.time_embed
                d_qpos_in = d_qpos
                d_qpos = zTD()
                gs_, ws_, ad_, dx_ = [d_q, d_k], [Wqp, Wkp], [d_qpos_in, None], [d_qpos, d_qpos]
                d_x = None
                if nd:
                    d_x = zTD()
                    gs_ += [d_q, d_k, d_v]; ws_ += [Wqc, Wkc, Wv]
                    ad_ += [ops._c(d_out_res.reshape(T, D)), None, None]; dx_ += [d_x, d_x, d_x]
                ops.linear_dgrad_multi(gs_, ws_, ad_, dx_, T, D, D)
                (dWqp, dbqp), (dWqt, dbqt), (dWqc, dbqc), (dWkp, dbkp) = zW(), zW(), zW(), zW()
                (dWkt, dbkt), (dWkc, dbkc), (dWv, dbv) = zW(), zW(), zW()
                _wgrad_multi([d_q, d_q, d_q, d_k, d_k, d_k, d_v], [qpos, te, x_out, qpos, te, x_out, x_out],
                             [dWqp, dWqt, dWqc, dWkp, dWkt, dWkc, dWv], [dbqp, dbqt, dbqc, dbkp, dbkt, dbkc, dbv], T, D, D)
            else:
                d_q, _, _, _ = _lin_b(r[0][0], q, W_in[:D], dw=dW_in[:D], db=dB_in[:D])
                d_k, _, _, _ = _lin_b(r[1][0], k, W_in[D:2 * D], dw=dW_in[D:2 * D], db=dB_in[D:2 * D])
                d_v, _, _, _ = _lin_b(r[2][0], v, W_in[2 * D:], dw=dW_in[2 * D:], db=dB_in[2 * D:])
                d_qpos, dWqp, dbqp, _ = _lin_b(d_q, qpos, Wqp, add=d_qpos)
                _, dWqt, dbqt, _ = _lin_b(d_q, ctx.time_embed, Wqt, need_dx=False)
                d_x, dWqc, dbqc, _ = _lin_b(d_q, x_out, Wqc, need_dx=nd, add=d_out_res.reshape(T, D) if nd else None)
                d_qpos, dWkp, dbkp, _ = _lin_b(d_k, qpos, Wkp, add=d_qpos)
                _, dWkt, dbkt, _ = _lin_b(d_k, ctx.time_embed, Wkt, need_dx=False)
                d_x, dWkc, dbkc, _ = _lin_b(d_k, x_out, Wkc, need_dx=nd, add=d_x)
                d_x, dWv, dbv, _ = _lin_b(d_v, x_out, Wv, need_dx=nd, add=d_x)
            # ---- query_scale / ref_point_head / sine embedding
            if not first:
                d_qsc = ops.ew2d(L.EW_MUL, d_sine_q, st["sine_h"])
                d_hs_, _, _, _ = _lin_b(d_qsc, st["x_s2"], Ws2, dw=dWs2, db=dbs2)
                d_x, _, _, _ = _lin_b(d_hs_, st["x_s1"], Ws1, dw=dWs1, db=dbs1, relu_y=st["h_s"], add=d_x)
            d_hr, _, _, _ = _lin_b(d_qpos, st["x_r2"], Wr2, dw=dWr2, db=dbr2)
            d_sine, _, _, _ = _lin_b(d_hr, st["x_r1"], Wr1, dw=dWr1, db=dbr1, relu_y=st["h_r"],
                                     need_dx=first and need_anchor)
            if first and need_anchor:
                ops.ew2d(L.EW_ADD, d_sine[:, :D], d_sine_q, out=d_sine[:, :D])
                d_a0 = ops.SineEmbedFn.backward(st["sine"], d_sine)
                d_anchor = _add(d_anchor, d_a0) if d_anchor is not None else d_a0
                if d_refs is not None:
                    d_anchor = _add(d_anchor, d_refs[0])
            d_next = d_x
            wgrad_flush(like)
            d_layers[i] = (dWqc, dbqc, dWqp, dbqp, dWqt, dbqt, dWkc, dbkc, dWkp, dbkp, dWkt, dbkt, dWv, dbv, dW_in, dB_in,
                           dWo, dbo, dg1, dbe1, dWcq, dbcq, dWcqp, dbcqp, dWqs, dbqs, dWo2, dbo2, dg3, dbe3, dW1, db1,
                           dW2, db2, dg4, dbe4, dWmk, dbmk, dWmp, dbmp, dWmv, dbmv)
        flat = tuple(d_sh)
        for t in d_layers:
            flat += t
        return ((d_mem.view(mshape) if d_mem is not None else None), None, None, d_anchor, None, None, None, None) + flat


def box_decoder(dec, memory, pos, kpm, anchor, time_embed):
    """dec: grounding.TransformerDecoder"""
    p = dec.layers[0].dropout_p if dec.training else 0.0
    wb = lambda m: (m.weight, m.bias)       # noqa: E731
    prm = (wb(dec.ref_point_head.layers[0]) + wb(dec.ref_point_head.layers[1]) + wb(dec.query_scale.layers[0])
           + wb(dec.query_scale.layers[1]) + wb(dec.bbox_embed.layers[0]) + wb(dec.bbox_embed.layers[1])
           + wb(dec.bbox_embed.layers[2]) + wb(dec.norm))
    for l in dec.layers:
        sa = l.self_attn
        prm += (wb(l.sa_qcontent_proj) + wb(l.sa_qpos_proj) + wb(l.sa_qtime_proj) + wb(l.sa_kcontent_proj)
                + wb(l.sa_kpos_proj) + wb(l.sa_ktime_proj) + wb(l.sa_v_proj)
                + (sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias) + wb(l.norm1)
                + wb(l.ca_qcontent_proj) + (wb(l.ca_qpos_proj) if l.ca_qpos_proj is not None else (None, None))
                + wb(l.ca_qpos_sine_proj) + wb(l.cross_attn.out_proj) + wb(l.norm3) + wb(l.linear1) + wb(l.linear2)
                + wb(l.norm4) + wb(l.ca_kcontent_proj) + wb(l.ca_kpos_proj) + wb(l.ca_v_proj))
    return plans.apply(BoxDecoderFn, memory, pos, kpm, anchor, time_embed, p, dec.layers[0].nhead, dec.num_layers, *prm)


# ------------------------------------------------------------------------------------------------------------------
# spatial-temporal encoder (modal_encoder.py:70-82, 104-204): token assembly + 2 x L layers as ONE node
# ------------------------------------------------------------------------------------------------------------------
_NE_LAYER = 12


def _cols(t3, r0, r1):
    """rows r0..r1 of every frame of a contiguous [n, S, d] tensor as a row-strided 2-D block [n, (r1-r0)*d]"""
    n, S, d = t3.shape
    return t3.view(n, S * d)[:, r0 * d:r1 * d]


class EncoderFn(Function):
    """inputs: vis_tokens [n,HW,d], txt [L,d], vis_pos [n,HW,d] (constant), kpm [n,1+HW+L] (bytes; column 0 = the frame
    [CLS] slot), tpos [1,n+1,d] (constant) -> (memory [n,HW+L,d], frames_cls [n,d], video_cls [1,d]).
    The reference concatenates / re-slices the token matrix around every layer (:145-151, :170-177, :195); here one
    buffer [n, 1+HW+L, d] is assembled once, the temporal layer's rows are written back into its [CLS] slots in place
    (as the reference's `src[0] = ...` does) and the backward routes those slots' gradients the same way."""

    @staticmethod
    def forward(ctx, vis_tokens, txt, vis_pos, kpm, tpos, p, nhead, nl, wpc, frame_cls, local_pos, video_cls, *prm):
        n, HW, d = vis_tokens.shape
        Lt = txt.shape[0]
        S1 = 1 + HW + Lt
        vis_tokens = vis_tokens if vis_tokens.is_contiguous() else vis_tokens.contiguous()
        txt = txt if txt.is_contiguous() else txt.contiguous()
        vis_pos = vis_pos if vis_pos.is_contiguous() else vis_pos.contiguous()
        x = ops._empty(vis_tokens, n, S1, d)
        ops.ew2d(L.EW_COPY, frame_cls.view(1, d), out=_cols(x, 0, 1))                          # :145-149
        ops.ew2d(L.EW_COPY, vis_tokens.view(n, HW * d), out=_cols(x, 1, 1 + HW))
        ops.ew2d(L.EW_COPY, txt.view(1, Lt * d), out=_cols(x, 1 + HW, S1))                      # :70-80
        pos = ops._zeros(vis_tokens, n, S1, d)                                                 # :82, :151
        ops.ew2d(L.EW_COPY, local_pos.view(1, d), out=_cols(pos, 0, 1))
        ops.ew2d(L.EW_COPY, vis_pos.view(n, HW * d), out=_cols(pos, 1, 1 + HW))
        video = video_cls.view(1, d)
        ctxs = []
        ffn_pl = None
        if (wpc is not None and FFN_PLANES and FUSE_FFN and L.get_mma_mode() == "bf16x6p" and n * S1 >= 4096 and d == 256
                and vis_tokens.is_cuda):
            # the spatial layers' FFN (13 248 x 256 -> 2048 at C3) on the plane kernels: ONE refresh of their weight planes
            pairs = [(prm[(2 * i) * _NE_LAYER + 6], prm[(2 * i) * _NE_LAYER + 8]) for i in range(nl)]
            need_bwd = any(w.requires_grad for pr in pairs for w in pr)
            wqks = [prm[(2 * i) * _NE_LAYER][:2 * d] for i in range(nl)] if QK_PLANES else []     # q / k rows of W_in
            ffn_pl = (wpc.refresh(pairs, need_bwd, extra=wqks), wpc)
        for i in range(nl):
            sp = prm[(2 * i) * _NE_LAYER:(2 * i + 1) * _NE_LAYER]
            tp = prm[(2 * i + 1) * _NE_LAYER:(2 * i + 2) * _NE_LAYER]
            c_s = _Ctx((True, True))
            x1 = EncoderLayerFn.forward(c_s, x, pos, kpm, p, nhead, *sp, ffn_pl=ffn_pl)        # :163-168
            seq = ops._empty(x, 1, n + 1, d)                                                   # :170-177
            ops.ew(L.EW_COPY, video, out=seq[0, 0:1])
            ops.ew2d(L.EW_COPY, _cols(x1, 0, 1), out=seq[0, 1:])
            c_t = _Ctx((True, False))
            seq2 = EncoderLayerFn.forward(c_t, seq, tpos, None, p, nhead, *tp)                 # :180-185
            video = seq2[0, 0:1]                                                               # :190
            ops.ew2d(L.EW_COPY, seq2[0, 1:], out=_cols(x1, 0, 1))                              # :195 (in place there too)
            x = x1
            ctxs.append((c_s, c_t))
        ctx.ctxs = ctxs
        ctx.dims = (n, HW, Lt, d, nl)
        memory = ops.ew2d(L.EW_COPY, _cols(x, 1, S1)).view(n, S1 - 1, d)
        frames = ops.ew2d(L.EW_COPY, _cols(x, 0, 1))
        return memory, frames, ops.ew(L.EW_COPY, video)

    @staticmethod
    def backward(ctx, d_memory, d_frames, d_video):
        like = d_memory if d_memory is not None else (d_frames if d_frames is not None else d_video)
        with wgrad_batch(like):
            return EncoderFn._backward(ctx, d_memory, d_frames, d_video)

    @staticmethod
    def _backward(ctx, d_memory, d_frames, d_video):
        n, HW, Lt, d, nl = ctx.dims
        S1 = 1 + HW + Lt
        like = d_memory if d_memory is not None else (d_frames if d_frames is not None else d_video)
        d_x = ops._zeros(like, n, S1, d) if (d_memory is None or d_frames is None) else ops._empty(like, n, S1, d)
        if d_memory is not None:
            ops.ew2d(L.EW_COPY, d_memory.contiguous().view(n, (S1 - 1) * d), out=_cols(d_x, 1, S1))
        if d_frames is not None:
            ops.ew2d(L.EW_COPY, d_frames.contiguous(), out=_cols(d_x, 0, 1))
        d_video = d_video.contiguous().view(1, d) if d_video is not None else ops._zeros(like, 1, d)
        d_lp = None                                   # [n,d] gradient rows of the [CLS] position embedding, all layers
        grads = [None] * (2 * nl)
        for i in reversed(range(nl)):
            c_s, c_t = ctx.ctxs[i]
            d_seq2 = ops._empty(like, 1, n + 1, d)
            ops.ew(L.EW_COPY, d_video, out=d_seq2[0, 0:1])
            ops.ew2d(L.EW_COPY, _cols(d_x, 0, 1), out=d_seq2[0, 1:])
            r = EncoderLayerFn.backward(c_t, d_seq2)          # (its own wgrad_batch: flushed per layer, joined at the end)
            grads[2 * i + 1] = r[5:]
            d_seq = r[0]
            d_video = d_seq[0, 0:1]
            ops.ew2d(L.EW_COPY, d_seq[0, 1:], out=_cols(d_x, 0, 1))       # the [CLS] slots of x1 fed the temporal layer
            r = EncoderLayerFn.backward(c_s, d_x)
            grads[2 * i] = r[5:]
            d_x, d_pos = r[0], r[1]
            if d_lp is None:
                d_lp = ops.ew2d(L.EW_COPY, _cols(d_pos, 0, 1))
            else:
                ops.ew2d(L.EW_ADD, d_lp, _cols(d_pos, 0, 1), out=d_lp)
        ni = ctx.needs_input_grad
        d_vis = ops.ew2d(L.EW_COPY, _cols(d_x, 1, 1 + HW)).view(n, HW, d) if ni[0] else None
        d_txt = None
        if ni[1]:
            d_txt = ops.colsum(ops.ew2d(L.EW_COPY, _cols(d_x, 1 + HW, S1))).view(Lt, d)
        d_fc = ops.colsum(ops.ew2d(L.EW_COPY, _cols(d_x, 0, 1))).view(1, d)
        flat = ()
        for g in grads:
            flat += tuple(g)
        ops.dropout_backward_done()   # (a frozen backbone has no backward node: this is then the path's last one)
        return (d_vis, d_txt, None, None, None, None, None, None, None, d_fc, ops.colsum(d_lp).view(1, d),
                ops.ew(L.EW_COPY, d_video)) + flat


def encoder(enc, vis_tokens, txt, vis_pos, kpm_full, tpos):
    """enc: grounding.SpatialTemporalEncoder"""
    p = enc.spatial_layers[0].dropout_p if enc.training else 0.0
    prm = ()
    for i in range(enc.num_layers):
        for l in (enc.spatial_layers[i], enc.temporal_layers[i]):
            sa = l.self_attn
            prm += (sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, l.norm1.weight, l.norm1.bias,
                    l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias, l.norm2.weight, l.norm2.bias)
    wpc = enc.__dict__.get("_ffn_planes")
    if wpc is None:
        wpc = enc.__dict__["_ffn_planes"] = FfnPlanes()
    return plans.apply(EncoderFn, vis_tokens, txt, vis_pos, kpm_full, tpos, p, enc.spatial_layers[0].nhead, enc.num_layers,
                           wpc, enc.frame_cls.weight, enc.local_pos_embed.weight, enc.video_cls.weight, *prm)


# ------------------------------------------------------------------------------------------------------------------
# span / actioness heads on the time decoder's states (pipeline.py:98-103): two MLPs with dropout, one node
# ------------------------------------------------------------------------------------------------------------------
class TimeHeadsFn(Function):
    """time_hs [L,T,D] -> (sted [L,T,2], act [L,T,1] | None); MLP = Linear-ReLU-dropout-Linear-dropout
    (net_utils.py:7-26 applies the dropout after every layer)"""

    @staticmethod
    def forward(ctx, time_hs, p, Wt1, bt1, Wt2, bt2, Wa1, ba1, Wa2, ba2):
        shp = time_hs.shape
        x = time_hs.contiguous().view(-1, shp[-1])
        st = []
        outs = []
        for (W1, b1, W2, b2) in ((Wt1, bt1, Wt2, bt2), (Wa1, ba1, Wa2, ba2)):
            if W1 is None:
                outs.append(None)
                st.append(None)
                continue
            h, x1 = _lin_f(x, W1, b1, relu=True)
            hd, c1 = (_f(ops.DropoutFn, _T, h, None, p) if p > 0.0 else (h, None))
            y, x2 = _lin_f(hd, W2, b2)
            yd, c2 = (_f(ops.DropoutFn, _T, y, None, p) if p > 0.0 else (y, None))
            outs.append(yd.view(*shp[:-1], W2.shape[0]))
            st.append((x1, h, c1, x2, c2, W1, W2))
        ctx.st = st
        ctx.shp = shp
        ctx.set_materialize_grads(False)
        if outs[1] is None:
            ctx.mark_non_differentiable()
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, d_sted, d_act):
        d_x = None
        grads = []
        for g, st in zip((d_sted, d_act), ctx.st):
            if st is None or g is None:
                grads += [None] * 4
                continue
            x1, h, c1, x2, c2, W1, W2 = st
            g = g.contiguous().view(-1, W2.shape[0])
            if c2 is not None:
                g = ops.DropoutFn.backward(c2, g)[0]
            d_hd, dW2, db2, _ = _lin_b(g, x2, W2)
            if c1 is not None:
                d_hd = ops.DropoutFn.backward(c1, d_hd.view(h.shape))[0]
            d_x, dW1, db1, _ = _lin_b(d_hd, x1, W1, relu_y=h, add=d_x)
            grads += [dW1, db1, dW2, db2]
        return (d_x.view(ctx.shp) if d_x is not None else None, None) + tuple(grads)


def time_heads(temp_embed, action_embed, time_hs):
    p = temp_embed.dropout_p if temp_embed.training else 0.0
    wb = lambda m: (m.weight, m.bias)       # noqa: E731
    a = (wb(action_embed.layers[0]) + wb(action_embed.layers[1])) if action_embed is not None else (None,) * 4
    return plans.apply(TimeHeadsFn, time_hs, p, *(wb(temp_embed.layers[0]) + wb(temp_embed.layers[1])), *a)
