"""Op-level parity of every C-ABI kernel against plain PyTorch fp32 CPU references.
Each body runs twice: on the host emulator (CPU, `-m "not gpu"`) and on the real
library (`-m gpu`).  Tolerances are fp32 round-off class (1e-4 relative to the
tensor scale; the north-star bar is 1e-3)."""
import math

import torch
import torch.nn.functional as F

from stcat_amd import _lib as L
from stcat_amd import ops
from tests.backends import both, close

TOL = 2e-4


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


def _linear_case(dev, M, N, K, relu, res, tile=None):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    r = rnd(M, N, seed=4) if res else None
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    rr = r.clone().requires_grad_(True) if res else None
    ref = F.linear(xr, wr, br)
    if res:
        ref = ref + rr
    gy = rnd(M, N, seed=5)
    if relu:
        # no upstream gradient where the pre-activation sits within round-off of the ReLU kink:
        # a flipped mask there is a measure-zero sensitivity, not a kernel property
        gy = gy * (ref.detach().abs() > 1e-3)
        ref = F.relu(ref)
    ref.backward(gy)
    if tile:
        L.call("stcat_debug_force_tile", *tile)
    try:
        xd, wd, bd = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
        rd = r.to(dev).requires_grad_(True) if res else None
        y = ops.linear(xd, wd, bd, rd, relu)
        y.backward(gy.to(dev))
    finally:
        L.call("stcat_debug_force_tile", 0, 0)
    tag = f"linear M{M} N{N} K{K} tile{tile}"
    close(y, ref, TOL, tag + " fwd")
    close(xd.grad, xr.grad, TOL, tag + " dx")
    close(wd.grad, wr.grad, TOL, tag + " dw")
    close(bd.grad, br.grad, TOL, tag + " db")
    if res:
        close(rd.grad, rr.grad, TOL, tag + " dres")


@both
def _linear(dev, big):
    _linear_case(dev, 70, 64, 64, relu=True, res=True)
    _linear_case(dev, 130, 128, 128, relu=False, res=False, tile=(128, 128))
    _linear_case(dev, 130, 128, 64, relu=True, res=True, tile=(128, 64))
    _linear_case(dev, 65, 192, 64, relu=False, res=True, tile=(64, 64))
    # decoder-sized launches (M <= 64 rows of frame queries), ragged M, fused bias + residual + ReLU
    _linear_case(dev, 64, 256, 256, relu=False, res=False)
    _linear_case(dev, 37, 64, 512, relu=True, res=True)
    _linear_case(dev, 8, 512, 64, relu=True, res=False)
    _linear_case(dev, 2, 128, 320, relu=False, res=True)
    if big:
        _linear_case(dev, 13248, 512, 256, relu=False, res=False)
        _linear_case(dev, 13248, 2048, 256, relu=True, res=False)
        _linear_case(dev, 13248, 256, 2048, relu=False, res=True)
        _linear_case(dev, 64, 256, 512, relu=True, res=False)


@both
def _small_linear(dev, big):
    for (M, N, K) in ((9, 4, 256), (7, 2, 256), (5, 1, 256)):
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
        xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
        ref = F.linear(xr, wr, br)
        gy = rnd(M, N, seed=5)
        ref.backward(gy)
        xd, wd, bd = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
        y = ops.linear(xd, wd, bd)
        y.backward(gy.to(dev))
        close(y, ref, TOL, f"small linear N{N}")
        close(xd.grad, xr.grad, TOL, "dx")
        close(wd.grad, wr.grad, TOL, "dw")
        close(bd.grad, br.grad, TOL, "db")


def _conv_case(dev, n, H, W, Cin, Cout, k, stride, pad, relu, res, tile=None):
    x = rnd(n, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    scale, bias = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    r = rnd(*ref.shape, seed=6) if res else None
    if res:
        ref = ref + r
    gy = rnd(*ref.shape, seed=5)
    if relu:
        gy = gy * (ref.detach().abs() > 1e-3)  # keep clear of the ReLU kink (see _linear_case)
        ref = F.relu(ref)
    ref.backward(gy)
    # NHWC / OHWI on the device
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    rd = r.permute(0, 2, 3, 1).contiguous().to(dev) if res else None
    sd, bd = scale.to(dev), bias.to(dev)
    if tile:
        L.call("stcat_debug_force_tile", *tile)
    try:
        y = ops.conv_fwd_raw(xd, wd, sd, bd, rd, stride, pad, relu)
        gyd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
        G, dres = ops.act_bwd_raw(gyd, y, sd, want_g=True, want_res=True, relu=relu)
        dx = ops.conv_dgrad_raw(G, wd, xd.shape, stride, pad)
        dw = ops.conv_wgrad_raw(G, xd, wd.shape, stride, pad)
        dx2 = ops.conv_dgrad_raw(G, wd, xd.shape, stride, pad, add=dx.clone())
        ymask = torch.randn_like(xd)
        msc = torch.rand(xd.shape[-1], device=dev) + 0.5
        dx3 = ops.conv_dgrad_raw(G, wd, xd.shape, stride, pad, mask_y=ymask, mask_scale=msc)
        dx4, dx5 = ops.conv_dgrad_raw(G, wd, xd.shape, stride, pad, add=dx.clone(), mask_y=ymask, scale2=msc)
    finally:
        L.call("stcat_debug_force_tile", 0, 0)
    tag = f"conv {k}x{k}/{stride} {Cin}->{Cout} {H}x{W} tile{tile}"
    close(y.permute(0, 3, 1, 2), ref, TOL, tag + " fwd")
    close(dx.permute(0, 3, 1, 2), xr.grad, TOL, tag + " dgrad")
    close(dx2.permute(0, 3, 1, 2), 2 * xr.grad, TOL, tag + " dgrad+add")
    ref3 = xr.grad.permute(0, 2, 3, 1) * (ymask.cpu() > 0) * msc.cpu()
    close(dx3, ref3, TOL, tag + " dgrad+fused relu/bn backward")
    ref4 = 2 * xr.grad.permute(0, 2, 3, 1) * (ymask.cpu() > 0)
    close(dx4, ref4, TOL, tag + " dgrad boundary dz")
    close(dx5, ref4 * msc.cpu(), TOL, tag + " dgrad boundary dz*scale")
    close(dw.permute(0, 3, 1, 2), wr.grad, TOL, tag + " wgrad")
    if res:
        mask = (ref > 0).float() if relu else torch.ones_like(ref)
        close(dres.permute(0, 3, 1, 2), gy * mask, TOL, tag + " dres")


@both
def _conv(dev, big):
    _conv_case(dev, 2, 7, 6, 64, 64, 3, 1, 1, relu=True, res=True)
    _conv_case(dev, 1, 8, 8, 64, 64, 3, 2, 1, relu=True, res=False)
    _conv_case(dev, 1, 9, 7, 64, 128, 1, 2, 0, relu=False, res=False)
    _conv_case(dev, 2, 5, 5, 128, 128, 1, 1, 0, relu=True, res=True, tile=(128, 128))
    _conv_case(dev, 1, 12, 11, 128, 128, 3, 1, 1, relu=True, res=False, tile=(128, 64))
    if big:
        _conv_case(dev, 4, 28, 28, 256, 256, 3, 1, 1, relu=True, res=False)
        _conv_case(dev, 4, 28, 28, 512, 256, 3, 2, 1, relu=True, res=False)
        _conv_case(dev, 4, 28, 28, 512, 1024, 1, 2, 0, relu=False, res=False)
        _conv_case(dev, 4, 14, 14, 1024, 256, 1, 1, 0, relu=True, res=True)
        _conv_case(dev, 8, 56, 56, 256, 128, 1, 1, 0, relu=True, res=False)


# ---------------------------------------------------------------------------------------
# plane-format conv family (mma mode "bf16x3p", csrc/igemm_pl.h): operands pre-split into bf16 hi/lo planes
# ---------------------------------------------------------------------------------------
def _pl_conv_case(dev, n, H, W, Cin, Cout, k, stride, pad, relu, res, tile=-1, wgrad=True, mode="bf16x3p"):
    """mode bf16x3p: two planes (16 significand bits, split error 2^-17); bf16x6p: three planes = the fp32 value exactly,
    six-term products (fp32-class: held to a 10x tighter bound than the 16-bit mode)"""
    old_mode = L.get_mma_mode()
    L.set_mma_mode(mode)
    if mode == "f16x3p":       # the test's gradients are O(1), not the 2^-15 of a training step: a loss scale of 2^2, not 2^16
        L.call("stcat_set_f16_scales", 6, 2)
    try:
        _pl_conv_body(dev, n, H, W, Cin, Cout, k, stride, pad, relu, res, tile, wgrad, mode)
    finally:
        L.call("stcat_set_f16_scales", 6, 16)
        L.set_mma_mode(old_mode)


def _pl_conv_body(dev, n, H, W, Cin, Cout, k, stride, pad, relu, res, tile, wgrad, mode):
    exact = mode == "bf16x6p"
    f16 = mode == "f16x3p"                # two fp16 planes: 22 significand bits (round trip 2^-23), three products
    TOL = 2e-5 if (exact or f16) else 2e-4
    RT = 0.0 if exact else (2e-6 if f16 else 2e-5)           # split / join round trip
    # mode f16x3p keeps gradient planes scaled by 2^glog and weight planes by 2^wlog (fp16's range): joined planes are
    # compared after undoing the power of two (exact)
    GS = L.f16_grad_scale()
    WS = float(2 ** L.load().stcat_get_f16_scale(0)) if f16 else 1.0
    unG = (lambda t: t / GS) if f16 else (lambda t: t)
    x = rnd(n, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    scale, bias = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=stride, padding=pad) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    r = rnd(*ref.shape, seed=6) if res else None
    if res:
        ref = ref + r
    gy = rnd(*ref.shape, seed=5)
    if relu:
        gy = gy * (ref.detach().abs() > 1e-3)
        ref = F.relu(ref)
    ref.backward(gy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    sd, bd = scale.to(dev), bias.to(dev)
    xp = ops.pl_split(xd)
    close(ops.pl_join(xp), xd, RT, "split/join round trip")  # hi + lo carries 16 significand bits
    rp = ops.pl_split(r.permute(0, 2, 3, 1).contiguous().to(dev)) if res else None
    cache = ops.WeightPlanes()
    wp, wt = cache.refresh([wd], transposed=True)
    wp, wt = wp[wd.data_ptr()], wt[wd.data_ptr()]
    close(ops.pl_join(wp) / WS, wd, RT, "weight planes")
    close(ops.pl_join(wt) / WS, wd.view(Cout, k * k, Cin).permute(1, 2, 0), RT, "transposed weight planes")
    L.call("stcat_debug_force_pl_tile", tile)
    try:
        yp, yf = ops.pl_conv_fwd_raw(xp, wp, sd, bd, rp, stride, pad, relu, planes_out=True, f32_out=True, want_mask=True)
        # the bit mask written by the epilogue == (y > 0), 8 columns per byte
        bits = ((yf.reshape(-1, Cout // 8, 8) > 0).to(torch.int32) << torch.arange(8, device=yf.device).to(torch.int32)).sum(-1)
        assert torch.equal(yp.mask.to(torch.int32), bits), "ReLU bit mask"
        gyd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
        G, dres = ops.pl_act_bwd_raw(gyd, yf, sd, want_g=True, want_res=True, relu=relu)
        dx = ops.pl_conv_dgrad_raw(G, wt, xd.shape, k, stride, pad)
        ymask = torch.randn_like(xd)
        if f16:      # fp16 planes flush |y| < 6e-8 to zero: the plane-derived ReLU mask (a path the model does not use: it
            ymask = torch.where(ymask.abs() < 1e-4, torch.full_like(ymask, 1e-4), ymask)   # reads bit masks) needs y off zero
        msc = torch.rand(xd.shape[-1], device=dev) + 0.5
        dx3 = ops.pl_conv_dgrad_raw(G, wt, xd.shape, k, stride, pad, mask_y=ops.pl_split(ymask), mask_scale=msc)
        ymp = ops.pl_split(ymask)
        ymp.mask = ((ymask.reshape(-1, Cin // 8, 8) > 0).to(torch.int32)
                    << torch.arange(8, device=ymask.device).to(torch.int32)).sum(-1).to(torch.uint8)
        dx4, dx5 = ops.pl_conv_dgrad_raw(G, wt, xd.shape, k, stride, pad, add=dx, mask_y=ymp, scale2=msc)  # bit-mask form
        dw = ops.pl_conv_wgrad_raw(G, xp, wd.shape, stride, pad) if wgrad else None
        if wgrad:
            # the workspace form (ordered sum of the slices' partial tiles: no atomics) is bit-identical run to run and agrees
            # with the atomic form (the same launch without a workspace)
            assert ops._wgrad_workspace(xd.device, L.stream_of(xd)) is not None
            assert torch.equal(dw, ops.pl_conv_wgrad_raw(G, xp, wd.shape, stride, pad)), "plane wgrad: not bit-reproducible"
            dw_at = ops._zeros(xd, *wd.shape)
            L.call("stcat_pl_conv_wgrad", G.h, G.l, xp.h, xp.l, dw_at.data_ptr(), None, n, H, W, Cin, Cout, k, k, stride, pad,
                   L.stream_of(xd))
            close(dw_at, dw, 5e-5, "plane wgrad: workspace form vs atomic form")
        gs = ops.pl_scale_raw(G, msc[:1].expand(Cout).contiguous())
        # FrozenBN scale folded into the transposed planes / the weight-gradient epilogue == running on g * scale
        fsc = torch.rand(Cout, device=dev) + 0.5
        _, wts = ops.WeightPlanes().refresh([wd], transposed=True, tscales=[fsc])
        Gs = ops.pl_scale_raw(G, fsc)
        dx_fold = ops.pl_conv_dgrad_raw(G, wts[wd.data_ptr()], xd.shape, k, stride, pad)
        dx_ref = ops.pl_conv_dgrad_raw(Gs, wt, xd.shape, k, stride, pad)
        dw_fold = ops.pl_conv_wgrad_raw(G, xp, wd.shape, stride, pad, row_scale=fsc).clone() if wgrad else None
        dw_sref = ops.pl_conv_wgrad_raw(Gs, xp, wd.shape, stride, pad).clone() if wgrad else None
    finally:
        L.call("stcat_debug_force_pl_tile", -1)
    tag = f"plane conv [{mode}] {k}x{k}/{stride} {Cin}->{Cout} {H}x{W} tile{tile}"
    close(yf.permute(0, 3, 1, 2), ref, TOL, tag + " fwd (fp32 out)")
    close(ops.pl_join(yp), yf, RT, tag + " fwd (planes out)")
    close(unG(ops.pl_join(dx)).permute(0, 3, 1, 2), xr.grad, TOL, tag + " dgrad")
    ref3 = xr.grad.permute(0, 2, 3, 1) * (ymask.cpu() > 0) * msc.cpu()
    close(unG(ops.pl_join(dx3)), ref3, TOL, tag + " dgrad+fused relu/bn backward")
    ref4 = 2 * xr.grad.permute(0, 2, 3, 1) * (ymask.cpu() > 0)
    close(unG(ops.pl_join(dx4)), ref4, TOL, tag + " dgrad boundary dz")
    close(unG(ops.pl_join(dx5)), ref4 * msc.cpu(), TOL, tag + " dgrad boundary dz*scale")
    close(unG(ops.pl_join(gs)), unG(ops.pl_join(G)).cpu() * msc[:1].cpu(), 2e-5, tag + " plane scale")
    close(unG(ops.pl_join(dx_fold)), unG(ops.pl_join(dx_ref)), 5e-5, tag + " dgrad with the scale folded into the weight planes")
    if wgrad:
        close(dw_fold, dw_sref, 5e-5, tag + " wgrad with the scale folded into the epilogue")
    if wgrad:
        close(dw.permute(0, 3, 1, 2), wr.grad, TOL, tag + " wgrad")
    if res:
        mask = (ref > 0).float() if relu else torch.ones_like(ref)
        close(unG(ops.pl_join(dres)).permute(0, 3, 1, 2), gy * mask, RT, tag + " dres")


@both
def _pl_conv(dev, big):
    # every tile shape of the table (0: 256x256, 1: 256x128, 2: 128x256, 3: 128x128, 4: 256x64); ragged row tails;
    # stride 2 (forward lattice + transposed-conv lattice); weight gradient needs Cin, Cout % 128 == 0
    _pl_conv_case(dev, 2, 7, 6, 64, 64, 3, 1, 1, relu=True, res=True, tile=4, wgrad=False)
    _pl_conv_case(dev, 1, 8, 8, 64, 128, 3, 2, 1, relu=True, res=False, tile=3, wgrad=False)
    _pl_conv_case(dev, 1, 9, 7, 128, 128, 1, 2, 0, relu=False, res=False, tile=1)
    # stride-2 data gradients on even dims run parity class by parity class (PlParams::par): 1x1 (three of the four
    # classes have no tap at all), padded 3x3 with a ragged class size, two frames
    _pl_conv_case(dev, 1, 10, 12, 128, 128, 1, 2, 0, relu=False, res=False, tile=3)
    _pl_conv_case(dev, 2, 6, 10, 64, 64, 3, 2, 1, relu=True, res=False, tile=4, wgrad=False)
    _pl_conv_case(dev, 2, 5, 5, 128, 256, 1, 1, 0, relu=True, res=True, tile=2)
    _pl_conv_case(dev, 1, 6, 5, 256, 256, 3, 1, 1, relu=True, res=False, tile=0)
    # three-plane mode (bf16x6p): its four tile shapes (1: 256x128, 2: 128x256, 3: 128x128, 4: 256x64), the parity-class
    # data gradient, residual planes, both wgrad tile families
    _pl_conv_case(dev, 2, 7, 6, 64, 64, 3, 1, 1, relu=True, res=True, tile=4, wgrad=False, mode="bf16x6p")
    _pl_conv_case(dev, 1, 9, 7, 128, 128, 1, 2, 0, relu=False, res=False, tile=1, mode="bf16x6p")
    _pl_conv_case(dev, 1, 10, 12, 128, 128, 1, 2, 0, relu=False, res=False, tile=3, mode="bf16x6p")
    _pl_conv_case(dev, 2, 5, 5, 128, 256, 1, 1, 0, relu=True, res=True, tile=2, mode="bf16x6p")
    _pl_conv_case(dev, 1, 6, 5, 256, 256, 3, 1, 1, relu=True, res=False, tile=0, mode="bf16x6p")
    # fp16-plane mode (f16x3p, round 4): every two-plane tile shape, the parity-class data gradient, residual planes, wgrad
    _pl_conv_case(dev, 2, 7, 6, 64, 64, 3, 1, 1, relu=True, res=True, tile=4, wgrad=False, mode="f16x3p")
    _pl_conv_case(dev, 1, 9, 7, 128, 128, 1, 2, 0, relu=False, res=False, tile=1, mode="f16x3p")
    _pl_conv_case(dev, 1, 10, 12, 128, 128, 1, 2, 0, relu=False, res=False, tile=3, mode="f16x3p")
    _pl_conv_case(dev, 2, 5, 5, 128, 256, 1, 1, 0, relu=True, res=True, tile=2, mode="f16x3p")
    _pl_conv_case(dev, 1, 6, 5, 256, 256, 3, 1, 1, relu=True, res=False, tile=0, mode="f16x3p")
    # tile 6: 128 x 64 with FOUR waves (two workgroups per CU; the K <= 512 1x1 convolutions of mode bf16x6p): residual,
    # ragged tail, the parity-class data gradient, 3x3 taps
    _pl_conv_case(dev, 2, 9, 7, 128, 128, 1, 1, 0, relu=True, res=True, tile=6, mode="bf16x6p")
    _pl_conv_case(dev, 1, 10, 12, 128, 128, 1, 2, 0, relu=False, res=False, tile=6, mode="bf16x6p")
    _pl_conv_case(dev, 2, 7, 6, 64, 64, 3, 1, 1, relu=True, res=True, tile=6, wgrad=False, mode="bf16x6p")
    # index 7: the A-stationary kernel (igemm_pl_as.h; three planes, 1x1 stride 1, K = 64 / 128 / 256, N >= 256 — forward
    # where Cin is the short side, data gradient where Cout is): activation rows in registers, weight ring of K / 32
    # slots, ragged last row block, residual planes, ReLU bit masks in and out, several units per row block
    _pl_conv_case(dev, 2, 13, 11, 256, 256, 1, 1, 0, relu=True, res=True, tile=7, mode="bf16x6p")
    _pl_conv_case(dev, 2, 9, 7, 128, 512, 1, 1, 0, relu=True, res=True, tile=7, mode="bf16x6p")
    _pl_conv_case(dev, 1, 12, 12, 64, 256, 1, 1, 0, relu=False, res=False, tile=7, wgrad=False, mode="bf16x6p")
    _pl_conv_case(dev, 3, 7, 7, 512, 128, 1, 1, 0, relu=True, res=False, tile=7, mode="bf16x6p")      # (data gradient: K = 128, N = 512)
    if big:
        _pl_conv_case(dev, 4, 28, 28, 256, 1024, 1, 1, 0, relu=True, res=True, mode="bf16x6p")           # layer3 conv3 / conv1 dgrad
        _pl_conv_case(dev, 4, 28, 28, 1024, 256, 1, 1, 0, relu=True, res=False, mode="bf16x6p")
        _pl_conv_case(dev, 2, 56, 56, 128, 512, 1, 1, 0, relu=True, res=True, mode="bf16x6p")
        for m3 in ("bf16x6p", "f16x3p"):
            _pl_conv_case(dev, 4, 28, 28, 256, 256, 3, 1, 1, relu=True, res=False, mode=m3)
            _pl_conv_case(dev, 4, 28, 28, 512, 256, 3, 2, 1, relu=True, res=False, mode=m3)
            _pl_conv_case(dev, 4, 28, 28, 512, 1024, 1, 2, 0, relu=False, res=False, mode=m3)
            _pl_conv_case(dev, 4, 14, 14, 1024, 256, 1, 1, 0, relu=True, res=True, mode=m3)
            _pl_conv_case(dev, 8, 56, 56, 256, 128, 1, 1, 0, relu=True, res=False, mode=m3)
            _pl_conv_case(dev, 8, 56, 56, 64, 64, 3, 1, 1, relu=True, res=False, wgrad=False, mode=m3)
            _pl_conv_case(dev, 16, 14, 14, 512, 512, 3, 1, 1, relu=True, res=False, mode=m3)
        _pl_conv_case(dev, 4, 28, 28, 256, 256, 3, 1, 1, relu=True, res=False)
        _pl_conv_case(dev, 4, 28, 28, 512, 256, 3, 2, 1, relu=True, res=False)
        _pl_conv_case(dev, 4, 28, 28, 512, 1024, 1, 2, 0, relu=False, res=False)
        _pl_conv_case(dev, 4, 14, 14, 1024, 256, 1, 1, 0, relu=True, res=True)
        _pl_conv_case(dev, 8, 56, 56, 256, 128, 1, 1, 0, relu=True, res=False)
        _pl_conv_case(dev, 8, 56, 56, 64, 64, 3, 1, 1, relu=True, res=False, wgrad=False)
        _pl_conv_case(dev, 16, 14, 14, 512, 512, 3, 1, 1, relu=True, res=False)


@both
def _pl_dgrad_coarse_add(dev, big):
    """stcat_pl_conv_dgrad_cadd: the block-boundary data gradient with the downsample branch's gradient taken from its own
    (stride-2) grid == the round-4 form that scatters it into a full-resolution tensor first; odd H / W, two and three planes"""
    old_mode = L.get_mma_mode()
    try:
        for mode in ("bf16x6p", "bf16x3p"):
            L.set_mma_mode(mode)
            for (n, H, W, Cin, Cout) in ((2, 9, 7, 128, 64), (1, 10, 12, 64, 128)) + (((4, 28, 28, 512, 256),) if big else ()):
                OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                g = rnd(n, H, W, Cout, seed=11).to(dev)                  # gradient at conv1's output
                w = rnd(Cout, 1, 1, Cin, seed=12, scale=Cin ** -0.5).to(dev)
                coarse = rnd(n, OH, OW, Cin, seed=13).to(dev)           # the downsample branch's gradient on its grid
                y = rnd(n, H, W, Cin, seed=14).to(dev)
                _, wt = ops.WeightPlanes().refresh([w], transposed=True)
                wt = wt[w.data_ptr()]
                gp, cp = ops.pl_split(g), ops.pl_split(coarse)
                yp = ops.pl_split(y)
                yp.mask = ((y.reshape(-1, Cin // 8, 8) > 0).to(torch.int32)
                           << torch.arange(8, device=y.device).to(torch.int32)).sum(-1).to(torch.uint8)
                full = torch.zeros(n, H, W, Cin, device=dev)
                full[:, ::2, ::2] = ops.pl_join(cp)
                want = ops.pl_conv_dgrad_raw(gp, wt, (n, H, W, Cin), 1, 1, 0, add=ops.pl_split(full), mask_y=yp)
                L.call("stcat_debug_force_pl_tile", 3 if not big else -1)
                try:
                    got = ops.pl_conv_dgrad_cadd_raw(gp, wt, (n, H, W, Cin), cp, 2, mask_y=yp)
                finally:
                    L.call("stcat_debug_force_pl_tile", -1)
                ref = (g.cpu().reshape(-1, Cout).double() @ w.cpu().reshape(Cout, Cin).double()).reshape(n, H, W, Cin)
                ref = (ref + full.cpu().double()) * (y.cpu() > 0)
                close(ops.pl_join(got), ref.float(), 2e-5 if mode == "bf16x6p" else 2e-4, f"coarse add [{mode}] vs fp64")
                close(ops.pl_join(got), ops.pl_join(want), 1e-6, f"coarse add [{mode}] vs scattered add")
    finally:
        L.set_mma_mode(old_mode)


@both
def _linear_multi(dev, big):
    """stcat_linear_{fwd,dgrad,wgrad}_multi: several skinny problems of one shape in one launch, outputs shared between
    problems (q = sum of three projections), against fp64"""
    old_mode = L.get_mma_mode()
    try:
        for mode in (("f32",) if dev.type == "cpu" else ()) + ("bf16x6", "bf16x3"):
            if mode == "f32":
                continue
            L.set_mma_mode(mode)
            tol = 2e-5 if mode == "bf16x6" else 3e-4
            for (M, N, K, n) in ((64, 256, 256, 7), (65, 128, 256, 3), (33, 256, 128, 2)):
                xs = [rnd(M, K, seed=10 + j).to(dev) for j in range(n)]
                ws = [rnd(N, K, seed=30 + j, scale=K ** -0.5).to(dev) for j in range(n)]
                bs = [rnd(N, seed=50 + j).to(dev) for j in range(n)]
                # outputs: problems 0..2 share y0 when n >= 3
                share = [0 if (n >= 3 and j < 3) else j for j in range(n)]
                ys = {k: torch.zeros(M, N, device=dev) for k in set(share)}
                ops.linear_fwd_multi(xs, ws, bs, [ys[share[j]] for j in range(n)], M, N, K)
                for k in ys:
                    want = sum(xs[j].cpu().double() @ ws[j].cpu().double().t() + bs[j].cpu().double() for j in range(n) if share[j] == k)
                    close(ys[k], want.float(), tol, f"linear_fwd_multi [{mode}] M{M} N{N} K{K} out{k}")
                gs = [rnd(M, N, seed=70 + j).to(dev) for j in range(n)]
                adds = [rnd(M, K, seed=90 + j).to(dev) if j % 2 == 0 else None for j in range(n)]
                dxs = {k: torch.zeros(M, K, device=dev) for k in set(share)}
                ops.linear_dgrad_multi(gs, ws, adds, [dxs[share[j]] for j in range(n)], M, N, K)
                for k in dxs:
                    want = sum(gs[j].cpu().double() @ ws[j].cpu().double() + (adds[j].cpu().double() if adds[j] is not None else 0)
                               for j in range(n) if share[j] == k)
                    close(dxs[k], want.float(), tol, f"linear_dgrad_multi [{mode}] out{k}")
                if N % 128 == 0 and K % 128 == 0:
                    dws = [torch.zeros(N, K, device=dev) for _ in range(n)]
                    dbs = [torch.zeros(N, device=dev) if j != 1 else None for j in range(n)]
                    ops.linear_wgrad_multi(gs, xs, dws, dbs, M, N, K)
                    for j in range(n):
                        close(dws[j], (gs[j].cpu().double().t() @ xs[j].cpu().double()).float(), tol, f"linear_wgrad_multi [{mode}] dw{j}")
                        if dbs[j] is not None:
                            close(dbs[j], gs[j].cpu().double().sum(0).float(), tol, f"linear_wgrad_multi [{mode}] db{j}")
    finally:
        L.set_mma_mode(old_mode)


@both
def _pl_linear_ffn(dev, big):
    """stcat_pl_linear_fwd / stcat_pl_linear_dgrad_mask (the encoder FFN's wide side on the plane kernels): y = dropout(relu(
    x W1^T + b)) with the kernels' own counter-based mask (== stcat_dropout on the dense tensor), its bit mask, and
    dx = [bits] gain (g W2) — against fp64"""
    old_mode = L.get_mma_mode()
    L.set_mma_mode("bf16x6p")
    try:
        for (M, K, N, p) in ((70, 256, 512, 0.25), (130, 128, 256, 0.0)) + (((13248, 256, 2048, 0.1),) if big else ()):
            x = rnd(M, K, seed=1).to(dev)
            w1 = rnd(N, K, seed=2, scale=K ** -0.5).to(dev)
            b1 = rnd(N, seed=3).to(dev)
            w2 = rnd(K, N, seed=4, scale=N ** -0.5).to(dev)             # linear2: [D, F]
            b2 = rnd(K, seed=7).to(dev)
            fwd, tr = ops.WeightPlanes().refresh([w1.view(N, 1, 1, K), w2.view(K, 1, 1, N)], transposed=True)
            w1p, w2t = fwd[w1.data_ptr()], tr[w2.data_ptr()]
            xa = rnd(M, K, seed=11).to(dev)                               # planes of a SUM in one pass (q = k = src + pos)
            sm, smp = ops.pl_split_sum(x, xa)
            assert torch.equal(sm, x + xa) and torch.equal(ops.pl_join(smp), sm), "pl_split_sum"
            xp = ops.pl_split(x)
            y = torch.empty(M, N, device=dev)
            bits = torch.empty(M, N // 8, device=dev, dtype=torch.uint8)
            ops.manual_seed(5)
            seed, off, base = ops._dropout_stream.take(M * N, x.device) if p > 0 else (0, 0, None)
            L.call("stcat_debug_force_pl_tile", 7 if not big else -1)
            try:
                yp = ops.Planes.empty(x, M, N)                             # fp32 AND plane output of the same launch
                L.call("stcat_pl_linear_fwd", xp.h, xp.l, w1p.h, w1p.l, b1.data_ptr(), None, y.data_ptr(), yp.h, yp.l,
                       bits.data_ptr(), M, N, K, 1, float(p), seed, off, base, L.stream_of(x))
                assert torch.equal(ops.pl_join(yp), y), "pl_linear_fwd: planes == fp32 result"
                ref = torch.relu(x.cpu().double() @ w1.cpu().double().t() + b1.cpu().double())
                if p > 0:
                    keep = torch.from_numpy(ops.dropout_keep_mask(seed, off + int(ops._dropout_stream.base(x.device).item()), M * N, p))
                    ref = ref * keep.view(M, N) / (1.0 - p)
                close(y, ref.float(), 2e-5, f"pl_linear_fwd M{M} K{K} N{N} p{p}")
                want_bits = ((y.reshape(M, N // 8, 8) > 0).to(torch.int32) << torch.arange(8, device=y.device).to(torch.int32)).sum(-1)
                assert torch.equal(bits.to(torch.int32), want_bits), "pl_linear_fwd bit mask"
                g = rnd(M, K, seed=6).to(dev)                              # gradient at linear2's output [M, D]
                gp = ops.pl_split(g)
                gain = 1.0 / (1.0 - p)
                gv = torch.full((N,), gain, device=dev)
                dx = torch.empty(M, N, device=dev)
                dxp = ops.Planes.empty(x, M, N)
                L.call("stcat_pl_linear_dgrad_mask", gp.h, gp.l, w2t.h, w2t.l, bits.data_ptr(), gv.data_ptr(), dx.data_ptr(),
                       dxp.h, dxp.l, M, K, N, L.stream_of(x))
                assert torch.equal(ops.pl_join(dxp), dx), "pl_linear_dgrad_mask: planes == fp32 result"
            finally:
                L.call("stcat_debug_force_pl_tile", -1)
            refd = (g.cpu().double() @ w2.cpu().double()) * (y.cpu() > 0) * gain
            close(dx, refd.float(), 2e-5, f"pl_linear_dgrad_mask M{M}")
            # the narrow side on the plane tile kernel: linear2 (K = N_wide -> D) with bias and an fp32 term, from the planes
            w2p, w1t = fwd[w2.data_ptr()], tr[w1.data_ptr()]
            addf = rnd(M, K, seed=8).to(dev)
            y2 = torch.empty(M, K, device=dev)
            L.call("stcat_pl_linear_fwd", yp.h, yp.l, w2p.h, w2p.l, b2.data_ptr(), addf.data_ptr(), y2.data_ptr(), None, None,
                   None, M, K, N, 0, 0.0, 0, 0, None, L.stream_of(x))
            ref2 = y.cpu().double() @ w2.cpu().double().t() + b2.cpu().double() + addf.cpu().double()
            close(y2, ref2.float(), 2e-5, f"pl_linear_fwd narrow M{M}")
            # linear1's data gradient from the plane gradient (transposed planes of W1), and both weight gradients + column sums
            dx1 = torch.empty(M, K, device=dev)
            L.call("stcat_pl_linear_fwd", dxp.h, dxp.l, w1t.h, w1t.l, None, None, dx1.data_ptr(), None, None, None,
                   M, K, N, 0, 0.0, 0, 0, None, L.stream_of(x))
            close(dx1, (dx.cpu().double() @ w1.cpu().double()).float(), 2e-5, f"linear1 dgrad on planes M{M}")
            if N % 128 == 0 and K % 128 == 0:
                v4 = lambda pl, C: ops.Planes(pl.t.view(pl.t.shape[0], 1, 1, M, C))   # noqa: E731
                dW2 = ops.pl_conv_wgrad_raw(v4(gp, K), v4(yp, N), (K, 1, 1, N), 1, 0).view(K, N)
                close(dW2, (g.cpu().double().t() @ y.cpu().double()).float(), 2e-5, f"linear2 wgrad on planes M{M}")
                dW1 = ops.pl_conv_wgrad_raw(v4(dxp, N), v4(xp, K), (N, 1, 1, K), 1, 0).view(N, K)
                close(dW1, (dx.cpu().double().t() @ x.cpu().double()).float(), 2e-5, f"linear1 wgrad on planes M{M}")
            close(ops.pl_colsum(dxp), dx.cpu().double().sum(0).float(), 2e-5, f"pl_colsum M{M}")
    finally:
        L.set_mma_mode(old_mode)


@both
def _pl_maxpool(dev, big):
    n, H, C = (2, 10, 64) if not big else (4, 112, 64)
    x = rnd(n, C, H, H, seed=1)
    ref = F.max_pool2d(x, 3, 2, 1)
    old_mode = L.get_mma_mode()
    try:
        for mode, tol in (("bf16x3p", 2e-5), ("bf16x6p", 0.0)):
            L.set_mma_mode(mode)
            y = ops.pl_maxpool_raw(x.permute(0, 2, 3, 1).contiguous().to(dev))
            assert y.t.shape[0] == (3 if mode == "bf16x6p" else 2)
            close(ops.pl_join(y).permute(0, 3, 1, 2), ref, tol, f"plane maxpool [{mode}]")
    finally:
        L.set_mma_mode(old_mode)


@both
def _stem_uint8_loader(dev, big):
    """uint8 HWC frames -> stem with ToTensor + Normalize fused into the gather == the fp32 path on the normalised
    NCHW tensor (datasets/vidstg.py:140, transforms.py:155-168), and == conv2d of the normalised tensor."""
    n, H, W = (2, 20, 26) if not big else (4, 224, 224)
    u8 = torch.randint(0, 256, (n, H, W, 3), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    mean, std = torch.tensor(ops.PIXEL_MEAN), torch.tensor(ops.PIXEL_STD)
    x = ((u8.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()       # ToTensor + Normalize, NCHW
    w = rnd(64, 3, 7, 7, seed=2, scale=147 ** -0.5)
    scale, bias = rnd(64, seed=3).abs() + 0.5, rnd(64, seed=4)
    ref = F.relu(F.conv2d(x, w, stride=2, padding=3) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    old_mode = L.get_mma_mode()
    try:
        for mode in (old_mode, "bf16x6p"):      # (bf16x6p: the LDS-staged six-product stem, csrc/stem_pl.h)
            L.set_mma_mode(mode)
            y8 = ops.stem_u8_fwd_raw(u8.to(dev), w.to(dev), scale.to(dev), bias.to(dev))
            yf = ops.stem_fwd_raw(x.to(dev), w.to(dev), scale.to(dev), bias.to(dev))
            close(y8.permute(0, 3, 1, 2), ref, TOL if mode != "bf16x6p" else 2e-5, f"uint8 stem vs conv2d of the normalised tensor ({mode})")
            close(y8, yf, 2e-5, f"uint8 stem vs fp32 stem ({mode})")
    finally:
        L.set_mma_mode(old_mode)


@both
def _stem_pool(dev, big):
    n, H = (2, 20) if not big else (4, 224)
    x = rnd(n, 3, H, H, seed=1)
    w = rnd(64, 3, 7, 7, seed=2, scale=147 ** -0.5)
    scale, bias = rnd(64, seed=3).abs() + 0.5, rnd(64, seed=4)
    ref = F.relu(F.conv2d(x, w, stride=2, padding=3) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
    old_mode = L.get_mma_mode()
    try:
        # the six-product stem (csrc/stem_pl.h: 16 x 16 output tiles, LDS-staged patch) on a square clip and on a non-square
        # one whose output (13 x 19) is not a multiple of the tile: fp32-class agreement with conv2d
        L.set_mma_mode("bf16x6p")
        y6 = ops.stem_fwd_raw(x.to(dev), w.to(dev), scale.to(dev), bias.to(dev))
        close(y6.permute(0, 3, 1, 2), ref, 2e-5, "six-product stem")
        xn = rnd(3, 3, 26, 38, seed=12) if not big else rnd(2, 3, 405, 720, seed=12)
        refn = F.relu(F.conv2d(xn, w, stride=2, padding=3) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1))
        yn = ops.stem_fwd_raw(xn.to(dev), w.to(dev), scale.to(dev), bias.to(dev))
        close(yn.permute(0, 3, 1, 2), refn, 2e-5, "six-product stem, non-square")
    finally:
        L.set_mma_mode(old_mode)
    y = ops.stem_fwd_raw(x.to(dev), w.to(dev), scale.to(dev), bias.to(dev))
    close(y.permute(0, 3, 1, 2), ref, TOL, "stem")
    p = ops.maxpool_raw(y)
    close(p.permute(0, 3, 1, 2), F.max_pool2d(ref, 3, 2, 1), TOL, "maxpool")
    wb, bb, rm, rv = rnd(64, seed=7), rnd(64, seed=8), rnd(64, seed=9), rnd(64, seed=10).abs() + 0.1
    s, b = ops.frozen_bn_fold(wb.to(dev), bb.to(dev), rm.to(dev), rv.to(dev))
    sr = wb * (rv + 1e-5).rsqrt()
    close(s, sr, 1e-6, "bn scale")
    close(b, bb - rm * sr, 1e-6, "bn bias")


@both
def _layernorm(dev, big):
    M = 37 if not big else 13248
    x, r = rnd(M, 256, seed=1, scale=3.0), rnd(M, 256, seed=2)
    g, b = rnd(256, seed=3) * 0.1 + 1, rnd(256, seed=4) * 0.1
    xr, rr, gr, br = [t.clone().requires_grad_(True) for t in (x, r, g, b)]
    ref = F.layer_norm(xr + rr, (256,), gr, br, 1e-5)
    gy = rnd(M, 256, seed=5)
    ref.backward(gy)
    xd, rd, gd, bd = [t.to(dev).requires_grad_(True) for t in (x, r, g, b)]
    y = ops.layer_norm(xd, gd, bd, res=rd)
    y.backward(gy.to(dev))
    close(y, ref, TOL, "ln fwd")
    close(xd.grad, xr.grad, TOL, "ln dx")
    close(rd.grad, rr.grad, TOL, "ln dres")
    close(gd.grad, gr.grad, TOL, "ln dgamma")
    close(bd.grad, br.grad, TOL, "ln dbeta")


def _mha_ref(q, k, v, kpm, scale, H):
    B, S, D = v.shape
    hd = D // H
    qh = (q * scale).view(B, S, H, hd).transpose(1, 2)
    kh = k.view(B, S, H, hd).transpose(1, 2)
    vh = v.view(B, S, H, hd).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2)
    if kpm is not None:
        sc = sc.masked_fill(kpm[:, None, None, :], float("-inf"))
    p = sc.softmax(-1)
    o = (p @ vh).transpose(1, 2).reshape(B, S, D)
    return o, p.mean(1)


def _mha_case(dev, B, S, H, need_w, packed, masked):
    D = H * 32
    qk = rnd(B, S, 2 * D, seed=1)
    v = rnd(B, S, D, seed=2)
    kpm = None
    if masked:
        kpm = torch.zeros(B, S, dtype=torch.bool)
        kpm[0, S - 3:] = True
        if B > 1:
            kpm[1, 1:4] = True
    scale = 32 ** -0.5
    qkr, vr = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    o_ref, w_ref = _mha_ref(qkr[..., :D], qkr[..., D:], vr, kpm, scale, H)
    go, gw = rnd(B, S, D, seed=3), rnd(B, S, S, seed=4)
    (o_ref * go).sum().backward(retain_graph=need_w)
    if need_w:
        (w_ref * gw).sum().backward()
    qkd, vd = qk.to(dev).requires_grad_(True), v.to(dev).requires_grad_(True)
    kd = kpm.to(dev) if masked else None
    if packed:
        o, w = ops.mha_self_packed(qkd, vd, kd, scale, need_w)
    else:
        o, w = ops.mha_self(qkd[..., :D], qkd[..., D:], vd, kd, scale, need_w)
    loss = (o * go.to(dev)).sum()
    if need_w:
        loss = loss + (w * gw.to(dev)).sum()
    loss.backward()
    tag = f"mha B{B} S{S} H{H} w{need_w} packed{packed}"
    close(o, o_ref, TOL, tag + " out")
    if need_w:
        close(w, w_ref, TOL, tag + " weights")
    close(qkd.grad, qkr.grad, TOL, tag + " dqk")
    close(vd.grad, vr.grad, TOL, tag + " dv")


@both
def _mha_self(dev, big):
    # inference path: no probability stash (pt = NULL) — same output as the training-path forward
    qk, v = rnd(2, 37, 128, seed=1).to(dev), rnd(2, 37, 64, seed=2).to(dev)
    with torch.no_grad():
        o_inf, _ = ops.mha_self_packed(qk, v, None, 32 ** -0.5)
    o_trn, _ = ops.mha_self_packed(qk.clone().requires_grad_(True), v, None, 32 ** -0.5)
    assert torch.equal(o_inf.cpu(), o_trn.detach().cpu())
    _mha_case(dev, 2, 37, 2, need_w=False, packed=True, masked=True)
    _mha_case(dev, 1, 8, 2, need_w=True, packed=True, masked=False)
    _mha_case(dev, 1, 65, 1, need_w=True, packed=False, masked=True)
    # the recomputing backward (opt-in, ops.MHA_RECOMPUTE: row max / sum kept, probability tiles rebuilt): masked, ragged,
    # unpacked, with dropout
    ops.MHA_RECOMPUTE = True
    try:
        _mha_case(dev, 2, 37, 2, need_w=False, packed=True, masked=True)
        _mha_case(dev, 1, 65, 1, need_w=False, packed=False, masked=True)
        _mha_dropout_case(dev, 2, 37, 2, need_w=False)
        if big:
            _mha_case(dev, 64, 207, 8, need_w=False, packed=True, masked=True)
            _mha_case(dev, 2, 256, 8, need_w=False, packed=True, masked=False)
            _mha_dropout_case(dev, 4, 207, 8, need_w=False, pdrop=0.1)
    finally:
        ops.MHA_RECOMPUTE = False
    # more than 256 tokens per frame (non-square clips: 405 x 720 -> 13 x 23 + text + [CLS] = 310): the long-row kernels
    # (K / V in dynamic LDS, two-pass softmax forward), forward + backward, masked, with and without the weights
    _mha_case(dev, 1, 310, 1, need_w=False, packed=True, masked=True)
    _mha_dropout_case(dev, 1, 270, 1, need_w=False)
    if big:
        _mha_case(dev, 8, 310, 8, need_w=False, packed=True, masked=True)
        _mha_case(dev, 2, 340, 8, need_w=True, packed=False, masked=True)
        _mha_case(dev, 2, 512, 8, need_w=False, packed=True, masked=False)
        _mha_case(dev, 2, 257, 8, need_w=False, packed=True, masked=True)
        _mha_case(dev, 64, 207, 8, need_w=False, packed=True, masked=True)
        _mha_case(dev, 3, 237, 8, need_w=False, packed=True, masked=True)
        _mha_case(dev, 1, 64, 8, need_w=True, packed=True, masked=False)
        for S in (1, 32, 33, 96, 129, 180, 256):
            _mha_case(dev, 2, S, 8, need_w=False, packed=True, masked=False)


def _q1_case(dev, B, S, H, two):
    D = H * 32
    q1, q2 = rnd(B, D, seed=1), rnd(B, D, seed=2)
    k1, k2, v = rnd(B, S, D, seed=3), rnd(B, S, D, seed=4), rnd(B, S, D, seed=5)
    kpm = torch.zeros(B, S, dtype=torch.bool)
    kpm[0, S // 2:] = True
    scale = (64 if two else 32) ** -0.5
    leaves = [t.clone().requires_grad_(True) for t in (q1, q2, k1, k2, v)]
    a1, a2, b1, b2, vv = leaves
    sc = (a1.view(B, 1, H, 32) * b1.view(B, S, H, 32)).sum(-1)
    if two:
        sc = sc + (a2.view(B, 1, H, 32) * b2.view(B, S, H, 32)).sum(-1)
    sc = (sc * scale).masked_fill(kpm[:, :, None], float("-inf"))
    p = sc.softmax(1)                                           # [B,S,H]
    ref = (p[..., None] * vv.view(B, S, H, 32)).sum(1).reshape(B, D)
    go = rnd(B, D, seed=6)
    ref.backward(go)
    dl = [t.to(dev).requires_grad_(True) for t in (q1, q2, k1, k2, v)]
    out = ops.attn_q1(dl[0], dl[1] if two else None, dl[2], dl[3] if two else None, dl[4], kpm.to(dev), scale)
    out.backward(go.to(dev))
    tag = f"q1 B{B} S{S} two{two}"
    close(out, ref, TOL, tag + " out")
    for i, nm in enumerate(("dq1", "dq2", "dk1", "dk2", "dv")):
        if not two and i in (1, 3):
            continue
        close(dl[i].grad, leaves[i].grad, TOL, tag + " " + nm)


@both
def _attn_q1(dev, big):
    _q1_case(dev, 3, 11, 2, True)
    _q1_case(dev, 5, 70, 1, False)
    _q1_case(dev, 2, 309, 1, True)      # two 256-key chunks per (frame, head): non-square clips
    if big:
        _q1_case(dev, 8, 309, 8, True)
        _q1_case(dev, 8, 339, 8, False)
        _q1_case(dev, 3, 700, 8, True)  # three chunks (the four-chunk instantiation)
        _q1_case(dev, 64, 206, 8, True)
        _q1_case(dev, 64, 206, 8, False)
        _q1_case(dev, 7, 256, 8, True)



# ---------------------------------------------------------------------------------------
# train-mode dropout: the kernels' counter-based masks replayed in the fp32 reference
# ---------------------------------------------------------------------------------------
def _mask_of(drop_p, seed, offset, n):
    import numpy as np
    keep = ops.dropout_keep_mask(seed, offset, n, drop_p)
    return torch.from_numpy(keep.astype(np.float32)) * float(np.float32(1.0 / (1.0 - float(np.float32(drop_p)))))


def _dropout_elementwise(dev, n, with_res):
    pdrop = 0.1
    x, r, g = rnd(n, seed=1), rnd(n, seed=2), rnd(n, seed=3)
    ops.manual_seed(1234, rank=3)
    seed, off = ops.dropout_stream_state()
    xd = x.to(dev).requires_grad_(True)
    rd = r.to(dev).requires_grad_(True)
    y = ops.dropout_add(xd, rd, pdrop) if with_res else ops.dropout(xd, pdrop)
    y.backward(g.to(dev))
    m = _mask_of(pdrop, seed, off, n)
    ref = x * m + (r if with_res else 0.0)
    assert torch.equal(y.detach().cpu(), ref), "dropout forward is not bit-identical to the host twin"
    assert torch.equal(xd.grad.cpu(), g * m)
    if with_res:
        assert torch.equal(rd.grad.cpu(), g)
    keep_rate = (m > 0).float().mean().item()
    assert abs(keep_rate - (1 - pdrop)) < 4.0 * math.sqrt(pdrop * (1 - pdrop) / n) + 1e-3, keep_rate
    # the stream advanced past this site: a second site gets a different mask
    assert ops.dropout_stream_state()[1] >= off + n
    y2 = ops.dropout(x.to(dev), pdrop)
    assert not torch.equal((y2 != 0).cpu(), (m > 0))
    # a new step restarts the host offset at 0 and advances the DEVICE base: same launch arguments, new mask
    ops.dropout_begin_step(dev)
    assert ops.dropout_stream_state()[1] == 0
    y3 = ops.dropout(x.to(dev), pdrop)
    m3 = _mask_of(pdrop, seed, ops._DropoutStream.STEP_SPAN, n)
    assert torch.equal(y3.cpu(), x * m3)


@both
def _dropout_stream_drop_in_mode(dev, big):
    """in drop-in mode nothing calls dropout_begin_step / manual_seed.  The stream must (a) open a new
    counter range by itself at the top of every train-mode forward, so the per-step range is never exhausted, and
    (b) seed every DP rank differently by default."""
    import os
    S = ops._DropoutStream
    ops.manual_seed(3)
    x = rnd(64, seed=1).to(dev)
    ops._dropout_stream.offset = S.STEP_SPAN - 64          # the host offset after ~190 C3 steps without a reset
    base0 = int(ops._dropout_stream.base(dev).item())
    ops.dropout_auto_begin_step(dev)                        # what Backbone.features_nhwc calls in train mode
    assert ops.dropout_stream_state()[1] == 0
    assert int(ops._dropout_stream.base(dev).item()) == base0 + S.STEP_SPAN
    ops.dropout_auto_begin_step(dev)                        # nothing drawn since: not doubled
    assert int(ops._dropout_stream.base(dev).item()) == base0 + S.STEP_SPAN
    y = ops.dropout(x, 0.5)
    assert ops.dropout_stream_state()[1] == 64 and torch.isfinite(y).all()
    # default seeds: a function of torch.initial_seed() and the rank
    seeds = []
    old = os.environ.get("RANK")
    try:
        for r in ("0", "1"):
            os.environ["RANK"] = r
            ops._dropout_stream.seed = None
            seeds.append(ops.dropout_stream_state()[0])
    finally:
        if old is None:
            os.environ.pop("RANK", None)
        else:
            os.environ["RANK"] = old
        ops.manual_seed(0)
    assert seeds[0] != seeds[1]


def _mha_dropout_case(dev, B, S, H, need_w, pdrop=0.25):
    D = H * 32
    SP = ((S + 31) // 32) * 32
    qk, v = rnd(B, S, 2 * D, seed=1), rnd(B, S, D, seed=2)
    scale = 32 ** -0.5
    ops.manual_seed(77)
    seed, off = ops.dropout_stream_state()
    # counter layout of the kernels: ((b*H + h)*SP + key)*SP + query
    m = _mask_of(pdrop, seed, off, B * H * SP * SP).view(B, H, SP, SP)[:, :, :S, :S].transpose(-1, -2)  # [B,H,q,k]
    qkr, vr = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    hd = 32
    qh = (qkr[..., :D] * scale).view(B, S, H, hd).transpose(1, 2)
    kh = qkr[..., D:].view(B, S, H, hd).transpose(1, 2)
    vh = vr.view(B, S, H, hd).transpose(1, 2)
    pd = (qh @ kh.transpose(-1, -2)).softmax(-1) * m
    o_ref = (pd @ vh).transpose(1, 2).reshape(B, S, D)
    w_ref = pd.mean(1)
    go, gw = rnd(B, S, D, seed=3), rnd(B, S, S, seed=4)
    loss = (o_ref * go).sum() + ((w_ref * gw).sum() if need_w else 0.0)
    loss.backward()
    qkd, vd = qk.to(dev).requires_grad_(True), v.to(dev).requires_grad_(True)
    o, w = ops.mha_self_packed(qkd, vd, None, scale, need_w, drop_p=pdrop)
    l2 = (o * go.to(dev)).sum()
    if need_w:
        l2 = l2 + (w * gw.to(dev)).sum()
    l2.backward()
    tag = f"mha-dropout B{B} S{S} H{H} w{need_w}"
    close(o, o_ref, TOL, tag + " out")
    if need_w:
        close(w, w_ref, TOL, tag + " weights")
    close(qkd.grad, qkr.grad, TOL, tag + " dqk")
    close(vd.grad, vr.grad, TOL, tag + " dv")


@both
def _mha_bs(dev, big):
    """bf16-pipe self-attention (csrc/attention_bs.h): online softmax, log-sum-exp stash, recompute backward."""
    L.set_mma_mode("bf16x3")
    try:
        _mha_case(dev, 2, 37, 2, need_w=False, packed=True, masked=True)       # two key tiles, ragged
        _mha_case(dev, 1, 65, 1, need_w=False, packed=False, masked=True)      # temporal-layer shape (T + 1)
        _mha_case(dev, 1, 150, 2, need_w=False, packed=True, masked=True)      # two 128-key chunks: online rescale
        _mha_dropout_case(dev, 2, 37, 2, need_w=False)
        # the rescale branch forced: one late key dominates every row (its chunk raises the running max by ~40)
        B, S, H, D = 1, 140, 1, 32
        qk, v = rnd(B, S, 2 * D, seed=11), rnd(B, S, D, seed=12)
        qk[0, :, :D] = qk[0, :, :D].abs() + 0.5
        qk[0, 135, D:] = 8.0
        o_ref, _ = _mha_ref(qk[..., :D], qk[..., D:], v, None, 32 ** -0.5, H)
        with torch.no_grad():
            o, _ = ops.mha_self_packed(qk.to(dev), v.to(dev), None, 32 ** -0.5)
        close(o, o_ref, TOL, "online-softmax rescale (late dominant key)")
        # more than 256 tokens per row (a 405 x 720 clip: 13 x 23 + 10 + 1 = 310): forward through key super-chunks
        B, S, H = 1, 310, 2
        qk, v = rnd(B, S, 2 * H * 32, seed=21), rnd(B, S, H * 32, seed=22)
        kpm = torch.zeros(B, S, dtype=torch.bool)
        kpm[0, 300:] = True
        o_ref, _ = _mha_ref(qk[..., :H * 32], qk[..., H * 32:], v, kpm, 32 ** -0.5, H)
        with torch.no_grad():
            o, _ = ops.mha_self_packed(qk.to(dev), v.to(dev), kpm.to(dev), 32 ** -0.5)
        close(o, o_ref, TOL, "S = 310 forward")
        # ... and with gradients: rows longer than 256 tokens take the fp32 long-row kernels in every mode
        _mha_case(dev, 1, 310, 1, need_w=False, packed=True, masked=True)
        if big:
            _mha_case(dev, 64, 207, 8, need_w=False, packed=True, masked=True)  # the C3 spatial layer
            _mha_dropout_case(dev, 4, 207, 8, need_w=False, pdrop=0.1)
    finally:
        L.set_mma_mode("f32")


@both
def _mha_bs_six_products(dev, big):
    """Round 6: mode bf16x6p's self-attention on the bf16 pipe (csrc/attention_bs.h, NP = 3): Q, K, V, the probabilities and
    dS as three bf16 planes (their sum IS the fp32 value), six products per contraction, fp32 accumulate; forward keeps the
    row log-sum-exp only, backward = two launches (dQ; dK / dV) that recompute the probabilities.  Checked against the
    fp32 PyTorch reference with the fp32 kernels' tolerance (TOL), and — what tells six products from three — much closer to
    an fp64 evaluation than the three-product kernel is."""
    L.set_mma_mode("bf16x6p")
    saved_min = ops.MHA_BS6_MIN_ROWS
    ops.MHA_BS6_MIN_ROWS = 0       # (the model routes rows of <= 128 tokens to the fp32-pipe kernels: here every shape runs NP = 3)
    try:
        _mha_case(dev, 2, 37, 2, need_w=False, packed=True, masked=True)       # two key tiles, ragged
        _mha_case(dev, 1, 65, 1, need_w=False, packed=False, masked=True)      # temporal-layer shape (T + 1)
        _mha_case(dev, 1, 150, 2, need_w=False, packed=True, masked=True)      # two 128-key chunks: online rescale
        _mha_case(dev, 1, 8, 2, need_w=True, packed=True, masked=False)        # head-mean weights wanted: fp32-pipe kernels
        _mha_dropout_case(dev, 2, 37, 2, need_w=False)
        _mha_case(dev, 1, 310, 1, need_w=False, packed=True, masked=True)      # > 256 tokens with gradients: long-row kernels
        # accuracy class: error against fp64, six products (this mode) vs three (mode bf16x3)
        B, S, H = 1, 96, 2
        D = H * 32
        qk, v, go = rnd(B, S, 2 * D, seed=31) * 2.0, rnd(B, S, D, seed=32), rnd(B, S, D, seed=33)
        qk64, v64 = qk.double().requires_grad_(True), v.double().requires_grad_(True)
        o64, _ = _mha_ref(qk64[..., :D], qk64[..., D:], v64, None, 32 ** -0.5, H)
        (o64 * go.double()).sum().backward()
        errs = {}
        for mode in ("bf16x6p", "bf16x3"):
            L.set_mma_mode(mode)
            a, b_ = qk.to(dev).requires_grad_(True), v.to(dev).requires_grad_(True)
            o, _ = ops.mha_self_packed(a, b_, None, 32 ** -0.5)
            (o * go.to(dev)).sum().backward()
            errs[mode] = tuple(((x.detach().double().cpu() - y.detach()).abs().max() / y.detach().abs().max()).item()
                               for x, y in ((o, o64), (a.grad, qk64.grad), (b_.grad, v64.grad)))
        L.set_mma_mode("bf16x6p")
        assert max(errs["bf16x6p"]) < 3e-6, errs                     # fp32-class (an fp32 evaluation itself: ~1e-6)
        assert max(errs["bf16x6p"]) < 0.2 * max(errs["bf16x3"]), errs
        if big:
            _mha_case(dev, 64, 207, 8, need_w=False, packed=True, masked=True)  # the C3 spatial layer
            _mha_case(dev, 2, 256, 8, need_w=False, packed=True, masked=False)
            _mha_dropout_case(dev, 4, 207, 8, need_w=False, pdrop=0.1)
    finally:
        ops.MHA_BS6_MIN_ROWS = saved_min
        L.set_mma_mode("f32")


def _q1_dropout_case(dev, B, S, H, pdrop=0.25):
    D = H * 32
    q1, k1, v = rnd(B, D, seed=1), rnd(B, S, D, seed=3), rnd(B, S, D, seed=5)
    scale = 32 ** -0.5
    ops.manual_seed(5)
    seed, off = ops.dropout_stream_state()
    m = _mask_of(pdrop, seed, off, B * H * S).view(B, H, S).transpose(1, 2)  # [B,S,H]
    a1, b1, vv = [t.clone().requires_grad_(True) for t in (q1, k1, v)]
    sc = (a1.view(B, 1, H, 32) * b1.view(B, S, H, 32)).sum(-1) * scale
    pd = sc.softmax(1) * m
    ref = (pd[..., None] * vv.view(B, S, H, 32)).sum(1).reshape(B, D)
    go = rnd(B, D, seed=6)
    ref.backward(go)
    dl = [t.to(dev).requires_grad_(True) for t in (q1, k1, v)]
    out = ops.attn_q1(dl[0], None, dl[1], None, dl[2], None, scale, drop_p=pdrop)
    out.backward(go.to(dev))
    tag = f"q1-dropout B{B} S{S}"
    close(out, ref, TOL, tag + " out")
    for t, r, nm in zip(dl, (a1, b1, vv), ("dq1", "dk1", "dv")):
        close(t.grad, r.grad, TOL, tag + " " + nm)


def _layernorm_dropout_case(dev, M, pdrop=0.2):
    x, r = rnd(M, 256, seed=1), rnd(M, 256, seed=2)
    gam, bet = rnd(256, seed=3) + 1.0, rnd(256, seed=4)
    gy = rnd(M, 256, seed=5)
    ops.manual_seed(99)
    seed, off = ops.dropout_stream_state()
    m = _mask_of(pdrop, seed, off, M * 256).view(M, 256)
    xr, rr, gr, br = [t.clone().requires_grad_(True) for t in (x, r, gam, bet)]
    ref = F.layer_norm(rr + xr * m, (256,), gr, br, 1e-5)
    ref.backward(gy)
    xd, rd, gd, bd = [t.to(dev).requires_grad_(True) for t in (x, r, gam, bet)]
    y = ops.layer_norm(xd, gd, bd, res=rd, drop_p=pdrop)
    y.backward(gy.to(dev))
    tag = f"layernorm+dropout M{M}"
    close(y, ref, TOL, tag + " y")
    close(xd.grad, xr.grad, TOL, tag + " dx")
    close(rd.grad, rr.grad, TOL, tag + " dres")
    close(gd.grad, gr.grad, TOL, tag + " dgamma")
    close(bd.grad, br.grad, TOL, tag + " dbeta")
    assert (xd.grad == 0).float().mean().item() > pdrop / 2      # dropped positions get no gradient


@both
def _dropout(dev, big):
    _layernorm_dropout_case(dev, 37)
    if big:
        _layernorm_dropout_case(dev, 13248)
    _dropout_elementwise(dev, 4096 + 3, with_res=True)
    _dropout_elementwise(dev, 1024, with_res=False)
    _mha_dropout_case(dev, 2, 37, 2, need_w=False)
    _mha_dropout_case(dev, 1, 40, 2, need_w=True)
    _q1_dropout_case(dev, 3, 21, 2)
    if big:
        _dropout_elementwise(dev, 13248 * 2048, with_res=False)
        _mha_dropout_case(dev, 16, 207, 8, need_w=False)
        _mha_dropout_case(dev, 1, 64, 8, need_w=True)
        _q1_dropout_case(dev, 64, 206, 8)

@both
def _elementwise(dev, big):
    a, b, c = rnd(6, 256, seed=1), rnd(6, 256, seed=2), rnd(6, 256, seed=3)
    leaves = [t.clone().requires_grad_(True) for t in (a, b, c)]
    dl = [t.to(dev).requires_grad_(True) for t in (a, b, c)]
    row = rnd(256, seed=4)
    ref = torch.tanh(torch.sigmoid(leaves[0] + leaves[1] + leaves[2]) * leaves[1] + row) + (leaves[0] + leaves[2])
    out = ops.add(ops.tanh(ops.add_const(ops.mul(ops.sigmoid(ops.add3(*dl)), dl[1]), row.to(dev))),
                  ops.add(dl[0], dl[2]))
    g = rnd(6, 256, seed=5)
    ref.backward(g)
    out.backward(g.to(dev))
    close(out, ref, TOL, "ew chain")
    for i in range(3):
        close(dl[i].grad, leaves[i].grad, TOL, f"ew grad {i}")
    # inverse sigmoid incl. clamp edges, and its gradient
    x = torch.tensor([-0.5, 0.0, 1e-4, 2e-3, 0.25, 0.5, 0.9, 0.9995, 1.0, 1.5] * 2).view(5, 4)
    xr = x.clone().requires_grad_(True)
    xc = xr.clamp(0, 1)
    ref = torch.log(xc.clamp(min=1e-3) / (1 - xc).clamp(min=1e-3))
    ref.sum().backward()
    xd = x.to(dev).requires_grad_(True)
    y = ops.inverse_sigmoid(xd)
    y.sum().backward()
    close(y, ref, 1e-5, "invsig")
    close(xd.grad, xr.grad, 1e-4, "invsig grad")
    # FiLM rows
    xx, gm, bt = rnd(7, 256, seed=6), rnd(256, seed=7), rnd(256, seed=8)
    lr = [t.clone().requires_grad_(True) for t in (xx, gm, bt)]
    ld = [t.to(dev).requires_grad_(True) for t in (xx, gm, bt)]
    gg = rnd(7, 256, seed=9)
    (lr[0] * lr[1] + lr[2]).backward(gg)
    ops.affine_rows(*ld).backward(gg.to(dev))
    for i in range(3):
        close(ld[i].grad, lr[i].grad, TOL, f"affine grad {i}")


@both
def _sine_and_pos(dev, big):
    from oracle import stcat_oracle as O
    anchors = torch.rand(9, 1, 4, generator=torch.Generator().manual_seed(3))
    ar = anchors.clone().requires_grad_(True)
    ref = O.gen_sineembed(ar)
    g = rnd(9, 1, 512, seed=1)
    ref.backward(g)
    ad = anchors.to(dev).requires_grad_(True)
    y = ops.sine_embed(ad)
    y.backward(g.to(dev))
    close(y, ref, 1e-5, "sine embed")
    close(ad.grad, ar.grad, 2e-4, "sine embed grad")
    m = torch.zeros(2, 5, 7, dtype=torch.bool)
    m[1, 3:, :] = True
    m[1, :, 5:] = True
    pos = ops.pos_sine_2d(m.to(dev))
    refp = O.pos_sine_2d(m).flatten(2).permute(0, 2, 1)
    close(pos, refp, 2e-5, "pos sine 2d")


@both
def _temporal_argmax(dev, big):
    from oracle import stcat_oracle as O
    for T, dur, seed in ((12, 12, 0), (12, 9, 1), (12, 6, 2), (64, 64, 3), (130, 100, 4), (5, 1, 5)):
        sted = rnd(1, T, 2, seed=seed) * 2
        if T == 12:
            sted[0, 3, 0] = sted[0, :, 0].max() + 1
            sted[0, 7, 1] = sted[0, :, 1].max() + 1
            sted[0, 9, 1] = sted[0, 7, 1]  # exact tie: first max must win
        boxes = torch.rand(T, 4)
        _, _, flat = O.post_process(sted, boxes, torch.ones(T, 2), list(range(T)), dur)
        got = ops.temporal_map_argmax(sted.to(dev), [dur]).cpu()
        assert (int(got[0, 0]), int(got[0, 1])) == (flat // T, flat % T), (T, dur, got, flat)


# ---- split-bf16 GEMM modes (stcat_set_mma_mode) ---------------------------------------------------
import contextlib


@contextlib.contextmanager
def mma_mode(mode, tol):
    global TOL
    old_tol, old_mode = TOL, L.get_mma_mode()
    L.set_mma_mode(mode)
    TOL = tol
    try:
        yield
    finally:
        L.set_mma_mode(old_mode)
        TOL = old_tol


def _gemm_family(dev, big):
    _linear_case(dev, 70, 64, 64, relu=True, res=True)
    _linear_case(dev, 130, 128, 128, relu=False, res=False, tile=(128, 128))
    _linear_case(dev, 130, 128, 64, relu=True, res=True, tile=(128, 64))
    _conv_case(dev, 2, 7, 6, 64, 64, 3, 1, 1, relu=True, res=True)
    _conv_case(dev, 1, 8, 8, 64, 64, 3, 2, 1, relu=True, res=False)
    _conv_case(dev, 1, 9, 7, 64, 128, 1, 2, 0, relu=False, res=False)
    _conv_case(dev, 2, 5, 5, 128, 128, 1, 1, 0, relu=True, res=True, tile=(128, 128))
    _conv_case(dev, 1, 12, 11, 128, 128, 3, 1, 1, relu=True, res=False, tile=(128, 64))
    if big:
        _linear_case(dev, 13248, 2048, 256, relu=True, res=False)
        _linear_case(dev, 13248, 256, 2048, relu=False, res=True)
        _conv_case(dev, 4, 28, 28, 256, 256, 3, 1, 1, relu=True, res=False)
        _conv_case(dev, 4, 28, 28, 512, 256, 3, 2, 1, relu=True, res=False)
        _conv_case(dev, 4, 14, 14, 1024, 256, 1, 1, 0, relu=True, res=True)


@both
def _gemm_bf16x3(dev, big):
    # products carry ~2^-16 relative error (two bf16 pieces per operand)
    with mma_mode("bf16x3", 1e-3):
        _gemm_family(dev, big)
        # the 8-wave 256x128 tile (one workgroup per CU): ragged M, 3x3 with padding, strided 1x1, residual + ReLU,
        # and the data gradient that runs through the same kernel on pre-transposed weights
        _linear_case(dev, 300, 128, 64, relu=True, res=True, tile=(256, 128))
        _conv_case(dev, 2, 13, 11, 64, 128, 3, 1, 1, relu=True, res=True, tile=(256, 128))
        _conv_case(dev, 1, 18, 17, 64, 256, 1, 2, 0, relu=False, res=False, tile=(256, 128))
        # ... and the 8-wave weight-gradient tile (Cout % 256 == 0, Cin % 128 == 0), 1x1 and padded 3x3, + bias sums
        _conv_case(dev, 2, 6, 7, 128, 256, 1, 1, 0, relu=True, res=False, tile=(256, 128))
        _conv_case(dev, 1, 5, 6, 128, 256, 3, 1, 1, relu=False, res=True, tile=(256, 128))
        _linear_case(dev, 100, 256, 128, relu=False, res=False, tile=(256, 128))
        # skinny long-reduction launches (decoder FFN): split-K over grid.z with an atomic epilogue — forward with
        # bias + residual (K = 2048) and the data gradient of a 256 -> 2048 layer (reduction over N = 2048)
        _linear_case(dev, 64, 256, 2048, relu=False, res=True)
        _linear_case(dev, 65, 256, 1024, relu=False, res=True)        # T+1 rows: two row tiles
        _linear_case(dev, 37, 2048, 256, relu=True, res=False)
        # stream-K scheduling of the forward GEMM (equal shares of tiles x K-steps per workgroup, split tiles
        # finished by the fix-up kernel): tile tails, whole tiles and tile heads; also the data gradient through it
        L.call("stcat_debug_streamk", 1)
        try:
            _linear_case(dev, 1000, 128, 128, relu=True, res=True)
            _conv_case(dev, 2, 23, 23, 64, 128, 3, 1, 1, relu=True, res=True)
            _conv_case(dev, 1, 40, 41, 64, 256, 1, 2, 0, relu=False, res=False)
            if big:
                _conv_case(dev, 64, 28, 28, 256, 256, 3, 1, 1, relu=True, res=False)
                _conv_case(dev, 37, 28, 28, 1024, 256, 1, 1, 0, relu=True, res=True)
        finally:
            L.call("stcat_debug_streamk", 0)
        if big:
            _conv_case(dev, 4, 28, 28, 256, 256, 3, 1, 1, relu=True, res=False, tile=(256, 128))
            _conv_case(dev, 4, 14, 14, 1024, 256, 1, 1, 0, relu=True, res=True, tile=(256, 128))


@both
def _gemm_bf16x6(dev, big):
    # three pieces per operand: fp32-class products
    with mma_mode("bf16x6", 2e-4):
        _gemm_family(dev, big)


@both
def _ew2d_strided(dev, big):
    """row-strided two-operand entry: column blocks of wider matrices in, column block out"""
    a, b = rnd(37, 96, seed=1), rnd(37, 64, seed=2)
    ad, bd = a.to(dev), b.to(dev)
    got = ops.ew2d(L.EW_MUL, ad[:, 32:64], bd[:, 16:48])
    close(got, a[:, 32:64] * b[:, 16:48], 1e-6, "ew2d mul")
    out = torch.zeros(37, 80, device=dev)
    ops.ew2d(L.EW_ADD, ad[:, :32], bd[:, 32:], out=out[:, 48:])
    close(out[:, 48:], a[:, :32] + b[:, 32:], 1e-6, "ew2d add into a column block")
    assert float(out[:, :48].abs().max()) == 0.0
    close(ops.ew2d(L.EW_COPY, ad[:, 5:69]), a[:, 5:69], 0.0, "ew2d copy")
    a3 = rnd(3, 5, 48, seed=3)
    close(ops.ew2d(L.EW_COPY, a3.to(dev)[..., 16:32]), a3[..., 16:32], 0.0, "ew2d copy 3-D")


def _stg_loss_case(dev, T, nl, s, e, seed, with_act=True, world_nb=None):
    """fused VideoSTGLoss kernel (values + gradients of every term and layer) vs the oracle's restatement of
    models/criterion.py evaluated layer by layer in float64 autograd"""
    from oracle import stcat_oracle as O
    from stcat_amd.misc import BoxList
    from stcat_amd.pipeline import VideoSTGLoss, weight_dict
    g = torch.Generator().manual_seed(seed)
    nbox = e - s + 1
    boxes = torch.rand(nl, T, 4, generator=g) * 0.5 + 0.2
    sted = torch.randn(nl, 1, T, 2, generator=g) * 2
    w = torch.softmax(torch.randn(nl, 1, T, T, generator=g), dim=-1)
    act = torch.randn(nl, 1, T, 1, generator=g)
    tgt = torch.rand(nbox, 4, generator=g) * 0.4 + 0.3
    actioness = torch.zeros(T, dtype=torch.bool)
    actioness[s:e + 1] = True
    wd = weight_dict(None, nl)
    # ---- oracle, float64
    ins64 = [t.double().requires_grad_(True) for t in (boxes, sted, w, act)]
    out = {"pred_boxes": ins64[0][nl - 1], "pred_sted": ins64[1][nl - 1], "weights": ins64[2][nl - 1],
           "pred_actioness": ins64[3][nl - 1],
           "aux_outputs": [{"pred_boxes": ins64[0][i], "pred_sted": ins64[1][i], "weights": ins64[2][i],
                            "pred_actioness": ins64[3][i]} for i in range(nl - 1)]}
    ref = O.criterion(out, actioness, tgt.double())
    if not with_act:
        ref = {k: v for k, v in ref.items() if "actioness" not in k}
    ref_total = sum(ref[k] * wd[k] for k in ref)
    ref_total.backward(retain_graph=True)
    # ---- HIP
    ins = [t.to(dev).requires_grad_(True) for t in (boxes, sted, w, act)]
    crit = VideoSTGLoss(None, losses=("boxes", "sted", "guided_attn") + (("actioness",) if with_act else ()))
    crit.weight_dict = wd
    outputs = {"pred_boxes": ins[0][nl - 1], "_stacked": {"pred_boxes": ins[0], "pred_sted": ins[1], "weights": ins[2],
                                                          "pred_actioness": ins[3]},
               "aux_outputs": [{} for _ in range(nl - 1)]}
    targets = [{"actioness": actioness.to(dev), "boxs": BoxList(tgt, (64, 64)).to(dev)}]
    losses = crit(outputs, targets, [T])
    assert set(losses) == set(ref), (sorted(losses), sorted(ref))
    for k in ref:
        close(losses[k], ref[k].float(), 2e-5, f"loss {k}")
    total = crit.weighted_total(wd)
    close(total, ref_total.float(), 2e-5, "weighted total")
    assert outputs["pred_boxes"].shape == (nbox, 4)           # criterion.py:168-171 side effect
    total.backward()
    for t, r, nm in zip(ins, ins64, ("boxes", "sted", "weights", "actioness")):
        if nm == "actioness" and not with_act:
            assert t.grad is None
            continue
        close(t.grad, r.grad.float(), 5e-5, f"d total / d {nm}")
    # the per-term route (sum of selected losses with other weights) goes through gvec
    for t in ins:
        t.grad = None
    losses = crit(outputs | {"pred_boxes": ins[0][nl - 1]}, targets, [T])
    (losses["loss_giou"] * 2.0 + losses["loss_sted_0"] * 3.0).backward()
    for r in ins64:
        r.grad = None
    (ref["loss_giou"] * 2.0 + ref["loss_sted_0"] * 3.0).backward()
    close(ins[0].grad, ins64[0].grad.float(), 5e-5, "d giou / d boxes")
    close(ins[1].grad, ins64[1].grad.float(), 5e-5, "d sted_0 / d sted")


@both
def _stg_loss(dev, big):
    _stg_loss_case(dev, 8, 3, 2, 5, seed=1)
    _stg_loss_case(dev, 5, 2, 0, 4, seed=2)                   # GT span = the whole clip: no negative rows
    _stg_loss_case(dev, 7, 6, 3, 3, seed=3, with_act=False)   # one-frame span, actioness head off
    if big:
        _stg_loss_case(dev, 64, 6, 10, 50, seed=4)
        _stg_loss_case(dev, 200, 6, 0, 120, seed=5)            # MAX_VIDEO_LEN


@both
def _linear_skinny_accumulate(dev, big):
    """the decoders' skinny launches with the step's zero arena on: outputs come zeroed from the arena and the reduction
    is split over grid.z (stcat_linear_fwd_acc / stcat_linear_dgrad_acc), forward with bias + residual, data gradient
    with the fused `add` operand"""
    from stcat_amd import composite
    arena = ops.enable_zero_arena(dev, 1 << 21)
    try:
        with mma_mode("bf16x3", 1e-3):
            for (M, N, K, res) in ((64, 256, 256, True), (65, 256, 2048, True), (37, 128, 128, False), (8, 2048, 256, False)):
                x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
                r = rnd(M, N, seed=4) if res else None
                used = arena.off
                y = ops.linear_fwd_raw(x.to(dev), w.to(dev), b.to(dev), r.to(dev) if res else None)
                assert arena.off > used, "the accumulate path did not take its output from the arena"
                ref = F.linear(x, w, b) + (r if res else 0)
                close(y, ref, TOL, f"skinny fwd {M}x{N}x{K}")
                g, add = rnd(M, N, seed=5), rnd(M, K, seed=6)
                dx, dw, db, _ = composite._lin_b(g.to(dev), x.to(dev), w.to(dev), add=add.to(dev))
                close(dx, g @ w + add, TOL, f"skinny dgrad {M}x{N}x{K}")
                close(dw, g.t() @ x, TOL, f"skinny wgrad {M}x{N}x{K}")
                close(db, g.sum(0), TOL, f"skinny bias grad {M}x{N}x{K}")
                arena.reset()
    finally:
        ops.disable_zero_arena()


@both
def _gemm_f32_lds_dma(dev, big):
    """exact-fp32 mode on the LDS-DMA kernel (igemm_pl_fwd_kernel<..., F32>): launches with >= 256 rows — padded 3x3,
    strided 3x3 and 1x1, residual + ReLU, ragged row tails, every tile shape, the data gradient on transposed weights
    with the fused mask / add operands, and Linear layers"""
    assert L.get_mma_mode() == "f32"
    _conv_case(dev, 2, 13, 11, 64, 128, 3, 1, 1, relu=True, res=True)          # M = 286
    _conv_case(dev, 1, 18, 17, 64, 256, 1, 2, 0, relu=False, res=False)
    _conv_case(dev, 1, 23, 21, 64, 64, 3, 2, 1, relu=True, res=False)
    _conv_case(dev, 1, 24, 22, 64, 64, 3, 2, 1, relu=True, res=False)           # even dims: parity-class data gradient
    _conv_case(dev, 1, 18, 18, 64, 128, 1, 2, 0, relu=False, res=False)
    _conv_case(dev, 3, 10, 10, 128, 64, 1, 1, 0, relu=True, res=True)
    _linear_case(dev, 300, 128, 64, relu=True, res=True)
    _linear_case(dev, 1000, 256, 128, relu=False, res=False)
    _linear_case(dev, 257, 64, 256, relu=False, res=True)
    for t in range(6):
        L.call("stcat_debug_force_pl_tile", t)
        try:
            _conv_case(dev, 1, 19, 17, 64, 256, 3, 1, 1, relu=True, res=True)
            _linear_case(dev, 515, 256, 64, relu=True, res=False)
        finally:
            L.call("stcat_debug_force_pl_tile", -1)
    if big:
        _conv_case(dev, 4, 28, 28, 256, 256, 3, 1, 1, relu=True, res=False)
        _conv_case(dev, 4, 28, 28, 512, 256, 3, 2, 1, relu=True, res=False)
        _conv_case(dev, 4, 14, 14, 1024, 256, 1, 1, 0, relu=True, res=True)
        _linear_case(dev, 13248, 2048, 256, relu=True, res=False)
        _linear_case(dev, 13248, 256, 2048, relu=False, res=True)
