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
            v.copy_(torch.from_numpy(synth.synth_value("map2d_head." + k, tuple(v.shape))))
    head.to(dev).eval()
    close(head.weight0, torch.from_numpy(g["map2d/weight0"]), 1e-6, "mask weight 0")
    close(head.weight1, torch.from_numpy(g["map2d/weight1"]), 1e-6, "mask weight 1")
    scores = head(torch.from_numpy(g["map2d/head/x"]).to(dev))
    close(scores, torch.from_numpy(g["map2d/head/scores"]), 1e-3, "TempPredictionHead scores", absolute=True)
    if big:  # the reference-sized head runs (128 x 128 map, 256 channels, four 9 x 9 convolutions)
        full = TempPredictionHead().to(dev).eval()
        out = full(torch.randn(1, 1, 64, 256, device=dev))
        assert out.shape == (1, 1, 128, 128) and torch.isfinite(out).all()
        assert torch.equal(out[0, 0] > 0, full.map_maker.mask2d) or (out[0, 0] * (~full.map_maker.mask2d)).abs().max() == 0
