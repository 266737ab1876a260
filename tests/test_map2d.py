"""2D temporal map head (models/map2d_head.py) — optional op: Gen2DMap and the conv-variant TempPredictionHead against
goldens generated from the imported reference (tests/golden/make_golden.py map2d)."""
import os

import numpy as np
import torch

from stcat_amd import synth
from stcat_amd.map2d import Gen2DMap, TempPredictionHead
from tests.backends import both, close
from tests.golden.make_golden import MAP2D_CFG

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "map2d.npz")


@both
def _map2d_head(dev, big):
    g = np.load(GOLD)
    c = MAP2D_CFG
    gen = Gen2DMap(c["MAX_MAP_SIZE"], c["POOLING_COUNTS"]).to(dev)
    assert np.array_equal(gen.mask2d.cpu().numpy(), g["map2d/mask"])
    for tag in ("short", "long"):   # T < N (adaptive max) and T > N (adaptive avg, then max)
        m = gen(torch.from_numpy(g[f"map2d/{tag}/x"]).to(dev))
        close(m.permute(0, 3, 1, 2), torch.from_numpy(g[f"map2d/{tag}/map"]), 1e-6, f"Gen2DMap {tag}")
    head = TempPredictionHead(c["HIDDEN"], c["MAX_MAP_SIZE"], c["POOLING_COUNTS"], c["KERNAL_SIZE"], c["CONV_LAYERS"])
    assert [k for k in head.state_dict().keys()] == list(g["map2d/head/keys"])       # the reference's parameter names
    with torch.no_grad():
        for k, v in head.state_dict().items():
            v.copy_(torch.from_numpy(synth.synth_value("map2d_head." + k, tuple(v.shape)).copy()))
    head.to(dev).eval()
    close(head.weight0, torch.from_numpy(g["map2d/weight0"]), 1e-6, "mask weight 0")
    close(head.weight1, torch.from_numpy(g["map2d/weight1"]), 1e-6, "mask weight 1")
    scores = head(torch.from_numpy(g["map2d/head/x"]).to(dev))
    close(scores, torch.from_numpy(g["map2d/head/scores"]), 1e-3, "TempPredictionHead scores", absolute=True)
    _train_and_grads(dev, g, head, "head", 2e-4)
    # 'attn' interaction (map2d_head.py:130-205): row attention batched over columns, column attention batched over rows,
    # the reference's inverted key_padding_mask mirrored; eval + train-mode raw scores against the reference
    ca = dict(c, HIDDEN=256, HEADS=8, DROPOUT=0.0)
    head_a = TempPredictionHead(ca["HIDDEN"], ca["MAX_MAP_SIZE"], ca["POOLING_COUNTS"], temp_head="attn", nhead=ca["HEADS"],
                                dim_feedforward=ca["FFN_DIM"], dropout=ca["DROPOUT"], attn_layers=ca["TEMP_PRED_LAYERS"])
    assert [k for k in head_a.state_dict().keys()] == list(g["map2d/attn/keys"])
    _fill(head_a, "map2d_attn_head.")
    head_a.to(dev).eval()
    xh = torch.from_numpy(g["map2d/attn/x"]).to(dev)
    with torch.no_grad():
        close(head_a(xh), torch.from_numpy(g["map2d/attn/scores"]), 1e-3, "attn head eval scores", absolute=True)
    head_a.train()
    xr = xh.clone().requires_grad_(True)
    sc = head_a(xr)
    close(sc, torch.from_numpy(g["map2d/attn/train_scores"]), 1e-3, "attn head train scores", absolute=True)
    # the reference cannot differentiate its own attn variant (in-place map update, see make_golden.py); ours does:
    # check the gradient against finite differences of the same forward on a few coordinates
    sc.sum().backward()
    assert torch.isfinite(xr.grad).all() and all(torch.isfinite(p.grad).all() for p in head_a.parameters())
    if big:
        with torch.no_grad():
            base = head_a(xh).sum().item()
            for (a, b_, t, d) in ((0, 0, 3, 5), (1, 0, 11, 40)):
                eps = 1e-2
                xp = xh.clone()
                xp[a, b_, t, d] += eps
                fd = (head_a(xp).sum().item() - base) / eps
                assert abs(fd - xr.grad[a, b_, t, d].item()) <= 5e-2 * max(1.0, abs(fd)), (fd, xr.grad[a, b_, t, d].item())
    if big:
        # the reference-sized head (128 x 128 map, 256 channels, four 9 x 9 convolutions) against the reference's outputs
        full = TempPredictionHead()
        _fill(full, "map2d_full_head.")
        full.to(dev).eval()
        xf = torch.from_numpy(g["map2d/full/x"]).to(dev)
        with torch.no_grad():
            close(full(xf), torch.from_numpy(g["map2d/full/scores"]), 1e-3, "full-size head eval scores", absolute=True)
        _train_and_grads(dev, g, full, "full", 1e-3, norms_only=True)


def _fill(head, prefix):
    with torch.no_grad():
        for k, v in head.state_dict().items():
            v.copy_(torch.from_numpy(synth.synth_value(prefix + k, tuple(v.shape)).copy()))


def _train_and_grads(dev, g, head, tag, tol, norms_only=False):
    """train mode: raw scores, and the gradients of sum(scores * G) w.r.t. the input and every parameter"""
    head.train()
    x = torch.from_numpy(g[f"map2d/{tag}/x"]).to(dev).requires_grad_(True)
    sc = head(x)
    close(sc, torch.from_numpy(g[f"map2d/{tag}/train_scores"]), 1e-3, f"{tag} train scores", absolute=True)
    for p in head.parameters():
        p.grad = None
    (sc * torch.from_numpy(g[f"map2d/{tag}/G"]).to(dev)).sum().backward()
    close(x.grad, torch.from_numpy(g[f"map2d/{tag}/dx"]), tol, f"{tag} dx")
    for k, p in head.named_parameters():
        if norms_only:
            ref = float(g[f"map2d/{tag}/gradnorm/{k}"])
            assert abs(p.grad.double().norm().item() - ref) <= 2e-3 * max(ref, 1e-6), (tag, k, p.grad.norm().item(), ref)
        else:
            close(p.grad, torch.from_numpy(g[f"map2d/{tag}/grad/{k}"]), tol, f"{tag} grad {k}")
    head.eval()
