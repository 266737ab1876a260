import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from backends import use_hip
from stcat_amd import _lib, ops, synth
from stcat_amd.misc import NestedTensor
from stcat_amd.pipeline import SyntheticText, build_model
dev = use_hip()
T, res, L = 8, 224, 10
for mode in ("bf16x3p", "f32"):
    _lib.set_mma_mode(mode)
    model, _, _ = build_model(None, SyntheticText(synth.synth_text(L)))
    model.eval(); synth.fill_module_(model); model.to(dev)
    u8 = torch.randint(0, 256, (T, res, res, 3), generator=torch.Generator().manual_seed(5), dtype=torch.uint8)
    mean, std = torch.tensor(ops.PIXEL_MEAN), torch.tensor(ops.PIXEL_STD)
    f32 = ((u8.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()
    mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
    with torch.no_grad():
        o8 = model(NestedTensor(u8.to(dev), mask, [T]), ["q"])
        of = model(NestedTensor(f32.to(dev), mask, [T]), ["q"])
        of2 = model(NestedTensor(f32.to(dev), mask, [T]), ["q"])
        f8 = model.vis_encoder[0].features_nhwc(u8.to(dev)); ff = model.vis_encoder[0].features_nhwc(f32.to(dev))
    for k in ("pred_boxes", "pred_sted", "pred_actioness"):
        print(mode, k, "u8 vs f32 %.3e" % (o8[k]-of[k]).abs().max().item(), " f32 vs f32 %.3e" % (of2[k]-of[k]).abs().max().item(), "scale %.3g" % of[k].abs().max().item())
    print(mode, "backbone features u8 vs f32: %.3e of %.3g" % ((f8-ff).abs().max().item(), ff.abs().max().item()))
