"""Host-side mirror of the reference module tree: state-dict keys/shapes, frozen set, factory seam,
boundary containers.  No kernels run (CPU)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import stcat_amd
from stcat_amd import synth
from stcat_amd.misc import BoxList, NestedTensor
from stcat_amd.pipeline import SyntheticText, build_model, weight_dict


@pytest.fixture(scope="module")
def model():
    m, crit, wd = build_model(None, SyntheticText(synth.synth_text(10)))
    return m, crit, wd


def test_state_dict_matches_reference_keys(model, golden_dir):
    m, _, _ = model
    ref_keys = [str(k) for k in np.load(os.path.join(golden_dir, "C1.npz"))["meta/state_dict_keys"]]
    sd = m.state_dict()
    assert set(sd.keys()) == set(ref_keys)
    shapes = dict(synth.hot_path_entries())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k


def test_frozen_parameters_match_backbonebase(model):
    m, _, _ = model
    for n, p in m.named_parameters():
        if n.startswith("vis_encoder."):
            want = any(s in n for s in ("layer2", "layer3", "layer4"))  # backbone.py:78-85
            assert p.requires_grad == want, n
        else:
            assert p.requires_grad, n
    conv = m.vis_encoder[0].body.layer3[0].conv2.weight
    assert conv.is_contiguous(memory_format=torch.channels_last)  # physically OHWI


def test_weight_dict_and_loss_keys(model):
    _, _, wd = model
    assert len(wd) == 30 and wd["loss_sted_4"] == 10.0 and wd["loss_giou"] == 3.0
    assert set(weight_dict()) == set(wd)


def test_install_rebinds_factory_seam():
    saved = {k: sys.modules.get(k) for k in ("models", "models.pipeline", "models.vision_model", "models.grounding_model")}
    try:
        for name in saved:
            sys.modules[name] = types.ModuleType(name)
        stcat_amd.install()
        from stcat_amd.backbone import build_vis_encoder
        from stcat_amd.grounding import build_decoder, build_encoder
        assert sys.modules["models.pipeline"].build_vis_encoder is build_vis_encoder
        assert sys.modules["models.pipeline"].build_encoder is build_encoder
        assert sys.modules["models.grounding_model"].build_decoder is build_decoder
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_unsupported_configurations_raise():
    from stcat_amd.backbone import Backbone, PositionEmbeddingSine
    with pytest.raises(ValueError):
        Backbone("resnet50")
    with pytest.raises(ValueError):
        PositionEmbeddingSine(64)


def test_nested_tensor_contract():
    a, b = torch.randn(3, 3, 4, 6), torch.randn(2, 3, 5, 4)
    nt = NestedTensor.from_tensor_list([a, b])
    assert nt.tensors.shape == (5, 3, 5, 6) and nt.durations == [3, 2]
    assert not nt.mask[0, :4, :6].any() and nt.mask[0, 4:].all() and nt.mask[4, :, 4:].all()
    sub = nt.subsample(2, 1)
    assert sub.durations == [1, 1] and torch.equal(sub.tensors[0], nt.tensors[1])
    assert len(BoxList(torch.zeros(7, 4))) == 7
    with pytest.raises(ValueError):
        BoxList(torch.zeros(7, 3))


def test_backbone_copies_without_its_staging_state():
    """round 6: the EMA copy of the training loop (copy.deepcopy(model), scripts/train_net.py:62-64) and pickling must not
    trip over the prefix pipeline's transient state (the declared frames, a HIP event, the resident plane buffers)"""
    import copy
    import pickle

    import torch

    from stcat_amd import backbone
    saved = backbone.BLOCKS
    backbone.BLOCKS = (1, 1, 1, 1)
    try:
        enc = backbone.build_vis_encoder(None)
    finally:
        backbone.BLOCKS = saved
    bb = enc[0]

    class _Unpicklable:
        def __reduce__(self):
            raise TypeError("cannot pickle")

    bb._staged = (torch.zeros(2), 0, _Unpicklable())
    bb._prefix = {"done": _Unpicklable(), "x": torch.zeros(3)}
    bb._pre_bufs = {"k": [torch.zeros(10)]}
    twin = copy.deepcopy(enc)
    assert twin[0]._staged is None and twin[0]._prefix is None and twin[0]._pre_bufs == {}
    assert bb._prefix is not None and bb._staged is not None                  # the original keeps running
    assert [n for n, _ in twin.named_parameters()] == [n for n, _ in enc.named_parameters()]
    pickle.dumps(enc)
