"""torch.ops.stcat_hip.*: the C ABI registered with the PyTorch dispatcher.  Every op resolves, and the
kernels are reachable THROUGH the dispatcher: results equal the fp32 PyTorch reference of the same op."""
import torch
import torch.nn.functional as F

import stcat_amd.torch_ops as T  # noqa: F401  (registers the namespace)
from tests.backends import both, close

S = torch.ops.stcat_hip


def test_namespace_resolves():
    names = T.registered_ops()
    assert len(names) >= 17
    for n in names:
        assert getattr(S, n) is not None, n
    for n in ("conv_bn_act_fwd", "conv_dgrad", "conv_wgrad", "maxpool3x3s2_fwd", "pos_sine_2d", "linear_bias_act_fwd",
              "linear_bwd", "layernorm_residual_fwd", "layernorm_residual_bwd", "mha_self_fwd", "mha_q1_cross_fwd",
              "sine_embed_anchor", "temporal_map_argmax"):
        assert n in names, n


@both
def _dispatcher_ops(dev, big):
    g = torch.Generator().manual_seed(0)
    # conv + FrozenBN + ReLU, its data / weight gradient
    n, H, Cin, Cout = 2, 6, 64, 64
    x = torch.randn(n, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5
    sc, bi = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, padding=1)
    y_ref = F.relu(ref * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1))
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xd, wd = x.permute(0, 2, 3, 1).contiguous().to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev)
    y = S.conv_bn_act_fwd(xd, wd, sc.to(dev), bi.to(dev), None, 1, 1, True)
    close(y.permute(0, 3, 1, 2), y_ref, 2e-4, "dispatcher conv_bn_act_fwd")
    gd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    close(S.conv_dgrad(gd, wd, list(xd.shape), 1, 1, None, None, None).permute(0, 3, 1, 2), xr.grad, 2e-4, "conv_dgrad")
    close(S.conv_wgrad(gd, xd, list(wd.shape), 1, 1).permute(0, 3, 1, 2), wr.grad, 2e-4, "conv_wgrad")
    # the plane-format form of the same conv: two bf16 planes per tensor (mode bf16x3p), three (mode bf16x6p)
    from stcat_amd import _lib
    for mode, np_ in (("bf16x3p", 2), ("bf16x6p", 3)):
        _lib.set_mma_mode(mode)
        try:
            xp = S.planes_split(xd)
            assert xp.shape[0] == np_ and xp.dtype == torch.bfloat16
            yp = S.conv_bn_act_fwd_planes(xp, S.planes_split(wd), sc.to(dev), bi.to(dev), None, 1, 1, True)
            close(S.planes_join(yp).permute(0, 3, 1, 2), y_ref, 2e-4, f"dispatcher conv_bn_act_fwd_planes {mode}")
        finally:
            _lib.set_mma_mode("f32")
    close(S.maxpool3x3s2_fwd(xd).permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2, 1), 1e-6, "maxpool")
    # linear (+ backward), LayerNorm (+ backward)
    M, K, N = 70, 256, 128
    a, wl, bl = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
    ar, wlr, blr = a.clone().requires_grad_(True), wl.clone().requires_grad_(True), bl.clone().requires_grad_(True)
    yl = F.linear(ar, wlr, blr)
    gl = torch.randn(M, N, generator=g)
    yl.backward(gl)
    close(S.linear_bias_act_fwd(a.to(dev), wl.to(dev), bl.to(dev), None, False), yl, 2e-4, "linear fwd")
    dx, dw, db = S.linear_bwd(gl.to(dev), a.to(dev), wl.to(dev), True)
    close(dx, ar.grad, 2e-4, "linear dx"); close(dw, wlr.grad, 2e-4, "linear dw"); close(db, blr.grad, 2e-4, "linear db")
    xx, rr = torch.randn(9, 256, generator=g), torch.randn(9, 256, generator=g)
    gam, bet = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    xxr = xx.clone().requires_grad_(True)
    ln = F.layer_norm(xxr + rr, (256,), gam, bet, 1e-5)
    gln = torch.randn(9, 256, generator=g)
    ln.backward(gln)
    yy, mean, rstd = S.layernorm_residual_fwd(xx.to(dev), rr.to(dev), gam.to(dev), bet.to(dev), 1e-5)
    close(yy, ln, 2e-5, "layernorm fwd")
    dz, _, _ = S.layernorm_residual_bwd(gln.to(dev), xx.to(dev), rr.to(dev), gam.to(dev), mean, rstd)
    close(dz, xxr.grad, 2e-5, "layernorm bwd")
    # attention cores, sine embedding, temporal map
    B, Sq, D = 2, 37, 256
    q, k, v = (torch.randn(B, Sq, D, generator=g) for _ in range(3))
    o, _ = S.mha_self_fwd(q.to(dev), k.to(dev), v.to(dev), None, 32 ** -0.5, False)
    qh, kh, vh = (t.view(B, Sq, 8, 32).transpose(1, 2) for t in (q, k, v))
    ref_o = (torch.softmax(qh @ kh.transpose(-1, -2) * 32 ** -0.5, -1) @ vh).transpose(1, 2).reshape(B, Sq, D)
    close(o, ref_o, 2e-4, "mha_self_fwd")
    q1 = torch.randn(B, D, generator=g)
    o1 = S.mha_q1_cross_fwd(q1.to(dev), None, k.to(dev), None, v.to(dev), None, 32 ** -0.5)
    p = torch.softmax(torch.einsum("bhd,bshd->bhs", q1.view(B, 8, 32), k.view(B, Sq, 8, 32)) * 32 ** -0.5, -1)
    close(o1, torch.einsum("bhs,bshd->bhd", p, v.view(B, Sq, 8, 32)).reshape(B, D), 2e-4, "mha_q1_cross_fwd")
    assert S.sine_embed_anchor(torch.rand(5, 4, generator=g).to(dev)).shape == (5, 512)
    sted = torch.randn(1, 12, 2, generator=g)
    out = S.temporal_map_argmax(sted.to(dev), [12]).cpu()
    ls, le = torch.log_softmax(sted[0, :, 0], 0), torch.log_softmax(sted[0, :, 1], 0)
    m = ls[:, None] + le[None, :] + torch.full((12, 12), -1e32).tril(0)
    idx = int(m.flatten().argmax())
    assert out.tolist() == [[idx // 12, idx % 12]]
