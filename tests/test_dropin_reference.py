"""Drop-in boundary against the REAL reference checkout (build container only; skipped where /root/reference is
absent, e.g. on the GPU box).  The reference's own models/pipeline.py builds STCATNet twice: with its own
factories, and after ``stcat_amd.install()`` with the HIP-backed modules (run here through the host emulator).
The second must strictly load the first's state dict and reproduce its outputs / losses / span."""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference checkout not present")


def test_install_is_a_drop_in_for_the_reference_pipeline():
    from tests.backends import close, use_emu
    from tests.golden import make_golden as G
    import stcat_amd
    from stcat_amd import synth

    T, res, L = 2, 64, 3
    cfg, ref_model, criterion, weight_dict, post, text = G.build_reference(L)
    from models import build_model          # reference package (stubs for absent third-party deps installed above)
    from utils.misc import NestedTensor as RefNested
    from utils.bounding_box import BoxList as RefBoxList
    import models.pipeline as ref_pipeline

    saved = (ref_pipeline.build_vis_encoder, ref_pipeline.build_encoder, ref_pipeline.build_decoder)
    try:
        stcat_amd.install()
        hip_model, _, _ = build_model(cfg)
    finally:
        (ref_pipeline.build_vis_encoder, ref_pipeline.build_encoder, ref_pipeline.build_decoder) = saved
    assert type(hip_model).__module__ == "models.pipeline"                      # the reference's own STCATNet class
    assert type(hip_model.ground_encoder).__module__.startswith("stcat_amd")
    missing, unexpected = hip_model.load_state_dict(ref_model.state_dict(), strict=True)  # checkpoint compatibility
    assert not missing and not unexpected
    hip_model.eval()

    use_emu()
    frames = synth.synth_frames(T, res)
    mask = torch.zeros(T, res, res, dtype=torch.bool)
    with torch.no_grad():
        out_ref = ref_model(RefNested(frames, mask.clone(), [T]), ["q"])
        out_hip = hip_model(RefNested(frames, mask.clone(), [T]), ["q"])
    for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights"):
        close(out_hip[k], out_ref[k], 1e-3, k)
        for a, b in zip(out_hip["aux_outputs"], out_ref["aux_outputs"]):
            close(a[k], b[k], 1e-3, "aux " + k)
    sizes = torch.tensor([[float(res), float(res)]]).repeat(T, 1)
    ids = [list(range(10, 10 + T))]
    b_ref, s_ref = post(out_ref, sizes, ids, [T])
    b_hip, s_hip = post(out_hip, sizes, ids, [T])
    assert s_ref == s_hip
    close(b_hip, b_ref, 1e-3, "post boxes")
    act, tb = synth.synth_targets(T)
    tg = lambda: [{"actioness": act, "boxs": RefBoxList(tb, (res, res), mode="xyxy")}]  # noqa: E731
    l_ref = criterion(out_ref, tg(), [T])
    l_hip = criterion(out_hip, tg(), [T])
    for k in l_ref:
        assert abs(l_ref[k].item() - l_hip[k].item()) <= 1e-3 * max(1.0, abs(l_ref[k].item())), k
