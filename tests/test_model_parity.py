"""Whole hot path (STCATNet forward, VideoSTGLoss, backward, PostProcess) through the C ABI against the reference.
Bars (BASELINE.json north_star): box / logit tensors within an ABSOLUTE 1e-3, argmax temporal span bit-exact, the 30 loss
terms, and every gradient tensor held to a bound calibrated with the reference's own fp64 run (class Ref, _compare).

* emulator variants (CPU, tiny clips, a shortened ResNet): the host wiring and every kernel's index logic end to end,
  against the CPU oracle run here — eval mode, train mode with the kernels' own dropout masks, padded / non-square clips.
* gpu variants: every model case of synth.MODEL_CASES at FULL size against fixtures the imported reference produced in
  the build container (tests/golden/model_*.npz: C1, C2, C3 = the benchmark's size, C5, padded and non-square clips), the
  benchmark's own step object replayed from its launch plans (C1, C3), and — round 6 — the TRAIN-mode step at C1 and C3
  against fixtures computed by the oracle from the recorded dropout stream of that very step (model_C{1,3}_train.npz).
"""
import contextlib
import math
import os

import numpy as np
import pytest
import torch

from oracle import stcat_oracle as O
from stcat_amd import synth
from stcat_amd.misc import BoxList, NestedTensor
from stcat_amd.pipeline import SyntheticText, build_model, build_postprocessors
from tests.backends import close, use_emu, use_hip

OUT_TOL = 1e-3
GRAD_TOL = 1e-3
BENCH_MMA = "bf16x6p"  # the arithmetic bench.py measures by default (three-plane backbone, fp32-class)
THROUGHPUT_MMA = "bf16x3p"  # bench.py's `throughput_mode` (two planes: 16 significand bits)
GRAD_ABS_FLOOR = 2e-6
# Calibrated gradient criterion of the fp32-class modes, tightened in round 4 with the measured distributions
# (profiles/r04_gpu_tests.log, `[gradient report]` lines): at C3 the worst tensor is 2.0e-3 from the fp64 run and 2 of 626
# tensors sit outside 3 x e_ref + 1e-3; at C1 / the 405 x 720 clip (8 frames: single ReLU-kink flips weigh 8x more) the worst
# is 1.2e-2 .. 2.8e-2 where the fp32 reference itself is 1e-3 .. 2.7e-2 away, 0-3 tensors outside (28 in the side mode bf16x6).
HARD_CAP = 2e-2          # EVERY gradient tensor: rel-L2 distance from exact arithmetic <= max(HARD_CAP, 3 x e_ref + 1e-3)  (was 5e-2)
OUTSIDE_FRACTION = 0.06  # share of the tensors that may sit outside the calibrated bound 3 x e_ref + 1e-3          (was 0.10)


_clip_of = synth.synth_clip     # frames [T,3,H,W] + padding mask [T,H,W] (pad="ragged": three partially padded frames)


# The emulator interprets every MFMA: a full ResNet-101 on a tiny clip is 3-6 minutes of CPU per test.  The emulator's
# model-level tests therefore run the same node code on a ResNet of 1 + 1 + 2 + 1 bottlenecks (frozen stem + layer1,
# trainable layer2-4, a stride-2 first block and a plain second block in layer3) — product AND oracle, same synthetic
# weights by name — except ONE test in the default arithmetic that keeps all 33 blocks; depth is covered on the GPU.
SMALL_NET = (1, 1, 2, 1)


@contextlib.contextmanager
def _depth(blocks):
    from stcat_amd import backbone
    saved = (backbone.BLOCKS, O.BLOCKS)
    if blocks is not None:
        backbone.BLOCKS = O.BLOCKS = tuple(blocks)
    try:
        yield
    finally:
        backbone.BLOCKS, O.BLOCKS = saved


def _run_hip(dev, T, res, L, with_backward=True, mma="f32", pad=None, graphed=False, blocks=None):
    from stcat_amd import _lib
    _lib.set_mma_mode(mma)
    try:
        with _depth(blocks):
            return _run_hip_impl(dev, T, res, L, with_backward, pad, graphed)
    finally:
        _lib.set_mma_mode("f32")


def _run_hip_impl(dev, T, res, L, with_backward, pad=None, graphed=False):
    text = synth.synth_text(L)
    model, criterion, wd = build_model(None, SyntheticText(text))
    model.eval()
    synth.fill_module_(model)
    model.to(dev)
    frames, mask, H, W = _clip_of(T, res, pad)
    frames, mask = frames.to(dev), mask.to(dev)
    if graphed:
        # decoder + heads through the hipGraph (STCATNet.capture_decoder, BASELINE configs[4]): captured on ANOTHER clip,
        # then replayed on this one — what comes out is a replay fed through the static input buffers
        assert not with_backward
        model.capture_decoder(True)
        with torch.no_grad():
            other = synth.synth_frames(T, max(H, W), seed=77)[:, :, :H, :W].contiguous().to(dev)
            model(NestedTensor(other, mask, [T]), ["synthetic"])
            out = model(NestedTensor(frames, mask, [T]), ["synthetic"])
        assert model._graphed_decoder.replays == 2 and len(model._graphed_decoder.graphs) == 1
    else:
        out = model(NestedTensor(frames, mask, [T]), ["synthetic"])
    keep = {k: v.detach().cpu().clone() for k, v in out.items() if torch.is_tensor(v)}
    keep["aux"] = [{k: v.detach().cpu().clone() for k, v in a.items()} for a in out["aux_outputs"]]
    sizes = torch.tensor([[float(H), float(W)]], device=dev).repeat(T, 1)
    boxes, sted = build_postprocessors()(out, sizes, [list(range(100, 100 + T))], [T])
    keep["post_boxes"], keep["post_sted"] = boxes.cpu(), sted
    losses = grads = None
    if with_backward or with_backward == "loss":
        act, tb = synth.synth_targets(T)
        targets = [{"actioness": act.to(dev), "boxs": BoxList(tb, (W, H)).to(dev)}]
        losses = criterion(out, targets, [T])
        total = sum(losses[k] * wd[k] for k in losses)
        if with_backward is True:
            total.backward()
            grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
        losses = {k: v.item() for k, v in losses.items()}
        losses["total"] = total.item()
    return keep, losses, grads


class Ref:
    """What a HIP run is compared with: the outputs / span / losses of the fp32 reference arithmetic and, per gradient
    tensor, a SAMPLE (synth.sample_indices: <= 1024 elements) of the fp32 gradient and of the same gradient computed in
    fp64 ("exact").  Two sources, one form:
      * `Ref.fixture(name)`: tests/golden/model_<name>.npz — the IMPORTED REFERENCE itself run in the build container
        (tests/golden/make_golden.py model ...), at every size the GPU tests use, C3 / C5 included: no full-size CPU run
        happens on the GPU box;
      * `Ref.oracle(...)`: the CPU oracle run here (the emulator's tiny clips)."""

    def __init__(self):
        self.out, self.aux, self.post_boxes, self.post_sted = {}, [], None, None
        self.losses = None
        self.grads = None        # name -> (sample of the fp32 gradient, sample of the fp64 gradient, numel): float64 arrays
        self.sampled = True      # fixtures keep synth.sample_indices elements per tensor; oracle runs keep every element
        self.dims = None         # (T, H, W, L, pad)

    @staticmethod
    def fixture(name) -> "Ref":
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"model_{name}.npz")
        g = np.load(path)
        r = Ref()
        T, H, W, L = (int(v) for v in g["meta/config"])
        r.dims = (T, H, W, L, str(g["meta/pad"]) or None)
        wT, wres, wL, wpad, _ = synth.MODEL_CASES[name]
        wH, wW = (wres, wres) if isinstance(wres, int) else wres
        assert r.dims == (wT, wH, wW, wL, wpad), (r.dims, synth.MODEL_CASES[name])     # fixture of THIS case definition
        keys = ("pred_boxes", "pred_sted", "pred_actioness", "weights")
        r.out = {k: torch.from_numpy(g[f"out/{k}"]) for k in keys}
        r.aux = [{k: torch.from_numpy(g[f"out/aux{i}/{k}"]) for k in keys} for i in range(5)]
        r.post_boxes = torch.from_numpy(g["post/boxes"])
        r.post_sted = g["post/sted"].tolist()
        if bool(g["meta/backward"]):
            r.losses = {str(k): float(v) for k, v in zip(g["loss/keys"], g["loss/values"])}
            r.losses["total"] = float(g["loss/total"])
            offs = g["grad/offsets"]
            s32, s64 = g["grad/sample32"].astype(np.float64), g["grad/sample64"].astype(np.float64)
            # (the reference's named_parameters() reports the shared box head under its first registration,
            # ground_decoder.decoder.bbox_embed.* — pipeline.py:50; canonical name: bbox_embed.*)
            r.grads = {synth.canonical_name(str(n)): (s32[offs[i]:offs[i + 1]], s64[offs[i]:offs[i + 1]],
                                                      int(g["grad/numel"][i])) for i, n in enumerate(g["grad/names"])}
        return r

    @staticmethod
    def oracle(T, res, L, with_backward=True, pad=None, sites=None, blocks=None) -> "Ref":
        """sites: a factory of oracle dropout-site objects (train mode with given masks; one object per run)"""
        with _depth(blocks):
            return Ref._oracle(T, res, L, with_backward, pad, sites)

    @staticmethod
    def _oracle(T, res, L, with_backward, pad, sites) -> "Ref":
        r = Ref()
        out, boxes, sted, losses, g32 = _run_oracle(T, res, L, with_backward, torch.float32, pad, sites)
        keys = ("pred_boxes", "pred_sted", "pred_actioness", "weights")
        r.out = {k: out[k].detach() for k in keys}
        r.aux = [{k: a[k].detach() for k in keys} for a in out["aux_outputs"]]
        r.post_boxes, r.post_sted, r.losses = boxes, [sted], losses
        if with_backward:
            g64 = _run_oracle(T, res, L, True, torch.float64, pad, sites)[4]
            r.sampled = False
            r.grads = {n: (g.reshape(-1).double().numpy(), g64[n].reshape(-1).double().numpy(), g.numel())
                       for n, g in g32.items()}
        return r


def _run_oracle(T, res, L, with_backward=True, dtype=torch.float32, pad=None, sites=None, trainable_only=False):
    """dtype=float64 gives the 'exact arithmetic' yardstick used to calibrate gradient tolerances.  trainable_only: only
    the parameters the reference trains require a gradient (stem / layer1 / FrozenBN buffers frozen, backbone.py:16-85) —
    same gradients for those, a fraction of the autograd memory (the full-size fixture runs, tests/golden/make_golden.py)."""
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        if sites is None:
            return _run_oracle_impl(T, res, L, with_backward, dtype, pad, trainable_only)
        with O.dropout_sites(sites()) as st:
            r = _run_oracle_impl(T, res, L, with_backward, dtype, pad, trainable_only)
        st.assert_all_consumed()
        return r
    finally:
        torch.set_default_dtype(prev)


_FROZEN = ("vis_encoder.0.body.conv1", "vis_encoder.0.body.bn1", "vis_encoder.0.body.layer1")


def _run_oracle_impl(T, res, L, with_backward, dtype, pad=None, trainable_only=False):
    sd = {k: v.to(dtype) for k, v in synth.synth_state_dict().items()}
    for k, v in sd.items():
        v.requires_grad_(not trainable_only or
                         not (k.startswith(_FROZEN) or ".bn" in k or "downsample.1" in k or k.endswith(".te")))
    frames, mask, H, W = _clip_of(T, res, pad)
    frames = frames.to(dtype)
    (tm, tmem, _), tcls = synth.synth_text(L)
    out = O.stcat_forward(sd, frames, mask, ((tm, tmem.to(dtype), None), tcls.to(dtype)))
    sizes = torch.tensor([[float(H), float(W)]]).repeat(T, 1)
    boxes, sted, _ = O.post_process(out["pred_sted"].detach(), out["pred_boxes"].detach(), sizes,
                                    list(range(100, 100 + T)), T)
    losses = grads = None
    if with_backward:
        act, tb = synth.synth_targets(T)
        l = O.criterion(out, act, tb.to(dtype))
        total = O.total_loss(l)
        total.backward()
        grads = {k: v.grad for k, v in sd.items() if v.grad is not None}
        losses = {k: v.item() for k, v in l.items()}
        losses["total"] = total.item()
    return out, boxes, sted, losses, grads


def _hip_case(dev, name, **kw):
    """the HIP run of a named model case (synth.MODEL_CASES)"""
    T, res, L, pad, bwd = synth.MODEL_CASES[name]
    kw.setdefault("with_backward", bwd)
    return _run_hip(dev, T, res, L, pad=pad, **kw)


def _family(name: str) -> str:
    for k in ("layer2", "layer3", "layer4"):
        if k in name:
            return "backbone." + k
    for k in ("input_proj", "ground_encoder", "ground_decoder.temp_decoder", "ground_decoder.decoder",
              "ground_decoder.template_generator", "temp_embed", "action_embed", "bbox_embed"):
        if name.startswith(k):
            return k
    return "other"


# Per-family caps on the rel-L2 distance of EVERY gradient tensor from exact (fp64) arithmetic in the 16-bit-operand
# modes (bf16x3 and its plane-format form bf16x3p): operands carry 16 significand bits (hi + lo bf16 pieces), so a
# product is good to ~2^-16 where fp32 gives 2^-24.  MEASURED on the GPU with tools/grad_error_report.py
# (profiles/r02_grad_error_{C1,C3}*.json).  Worst tensor per family, bf16x3 | bf16x3p:
#   C3 (T=64, 448^2, the benchmark): layer2 7.2e-3 | 6.7e-3, layer3 4.1e-3 | 4.2e-3, layer4 1.9e-3 | 1.8e-3,
#      input_proj 1.0e-3 | 5.5e-4, encoder 1.0e-3 | 8.4e-4, box decoder 2.4e-3 | 4.1e-3, time decoder 3.4e-3 | 1.3e-3,
#      temp_embed 9.1e-3 | 9.1e-3  (exact-fp32 mode: layer2 1.3e-3; the fp32 CPU reference itself: 1.5e-3);
#   C1 (T=8, 224^2: 8x fewer samples per gradient, a flipped ReLU kink weighs more): layer2 8.0e-3 | 1.4e-2,
#      layer3 7.7e-3 | 1.1e-2, layer4 1.7e-3 | 6.5e-3, input_proj 6.6e-4 | 2.7e-3, encoder 5.2e-4 | 2.4e-3
#      (the fp32 CPU reference: layer2 1.2e-2, layer3 2.5e-2).
# Caps = the larger measurement x ~1.5; none above 2e-2 (the former 20x slack allowed 0.2).
# A tensor whose fp32-reference gradient is itself far from exact (an ill-conditioned tensor: one ReLU kink of a tiny
# MLP) is allowed 2 x that distance + 1e-3 instead.
GRAD_CAPS_16BIT = {
    "backbone.layer2": 2e-2, "backbone.layer3": 1.6e-2, "backbone.layer4": 1e-2, "input_proj": 4e-3,
    "ground_encoder": 4e-3, "ground_decoder.temp_decoder": 6e-3, "ground_decoder.decoder": 6e-3,
    "ground_decoder.template_generator": 2e-3, "temp_embed": 1.5e-2, "action_embed": 2e-3, "bbox_embed": 2e-3,
    "other": 2e-3,
}


def _compare(hip, ref: Ref, with_backward=True, grad_slack=1.0, grad_caps=None, report_to=None):
    """Outputs / spans / losses always use the north-star bars.  Gradients: fp32-class modes (f32, bf16x6, bf16x6p) must
    be as close to exact arithmetic as the fp32 reference is (calibrated bound below, grad_slack = 1); the
    16-bit-operand modes pass `grad_caps` = per-family caps on every tensor's rel-L2 error.  Gradient errors are
    measured on the reference's per-tensor sample (class Ref); report_to: a list that receives the per-tensor rows."""
    keep, losses, grads = hip
    for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights"):
        close(keep[k], ref.out[k], OUT_TOL, k, absolute=True)       # "box/logit tensors within 1e-3": absolute
        for i, aux in enumerate(keep["aux"]):
            close(aux[k], ref.aux[i][k], OUT_TOL, f"aux{i}/{k}", absolute=True)
    close(keep["post_boxes"], ref.post_boxes, OUT_TOL, "post boxes")
    assert keep["post_sted"] == ref.post_sted, (keep["post_sted"], ref.post_sted)  # bit-exact span
    if not with_backward:
        return
    for k, v in ref.losses.items():
        assert abs(losses[k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses[k], v)
    missing, report = [], []
    seen = set()
    for name, (g32, g64, numel) in ref.grads.items():
        if name.startswith("ground_decoder.decoder.bbox_embed."):
            continue  # alias of bbox_embed.* (pipeline.py:50)
        if name.endswith(".te"):
            continue  # sine time table: a buffer
        if name.startswith("vis_encoder.") and (
                not any(s in name for s in ("layer2", "layer3", "layer4"))   # frozen: backbone.py:78-85
                or ".bn" in name or "downsample.1" in name):                 # FrozenBN buffers
            continue
        hip_name = name
        if name.startswith("bbox_embed."):
            hip_name = "ground_decoder.decoder." + name  # named_parameters() reports the first registration
        if hip_name not in grads:
            if float(np.abs(g32).max()) > 0:
                missing.append(name)
            continue
        seen.add(hip_name)
        # Yardstick = the same gradient computed in fp64 ("exact").  The fp32 reference itself is only
        # conditioned to a few 1e-3 on some tensors of this 104-conv + 18-layer chain (ReLU-kink flips,
        # softmax/LayerNorm amplification), so the HIP path is required to be as close to exact
        # arithmetic as the fp32 reference is (x3 + 1e-3), per tensor, in relative L2.  Absolute floor:
        # key-side attention biases have an exactly-zero gradient (softmax shift invariance).
        assert grads[hip_name].numel() == numel, (name, grads[hip_name].shape, numel)
        a = grads[hip_name].reshape(-1)
        if ref.sampled:
            a = a[torch.from_numpy(synth.sample_indices(name, numel))]
        a = a.double().numpy()
        floor = GRAD_ABS_FLOOR * g64.size ** 0.5 / GRAD_TOL
        nrm = float(np.linalg.norm(g64)) + floor
        e_hip = float(np.linalg.norm(a - g64)) / nrm
        e_ref = float(np.linalg.norm(g32 - g64)) / nrm
        gross = float(np.abs(a - g64).max()) / (float(np.abs(g64).max()) + GRAD_ABS_FLOOR / GRAD_TOL)
        report.append((e_hip / (3 * e_ref + GRAD_TOL * grad_slack), e_hip, e_ref, gross, name))
    if report_to is not None:
        report_to.extend(report)
    _print_gradient_report(os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0], report)
    assert not missing, f"parameters without a HIP gradient: {missing[:8]}"
    # parameters that get no gradient in the reference (fusion, ca_qtime_proj) get none here
    for name in grads:
        ref_name = name.replace("ground_decoder.decoder.bbox_embed.", "bbox_embed.")
        assert name in seen or ref_name in ref.grads, f"unexpected gradient for {name}"
    if grad_caps is not None:
        cap_of = lambda n, r: max(grad_caps[_family(n)], 2 * r + GRAD_TOL)  # noqa: E731
        over = [(h / cap_of(n, r), h, r, g, n) for _, h, r, g, n in report if h > cap_of(n, r)]
        over.sort(reverse=True)
        assert not over, "gradient rel-L2 error above the family cap: " + "; ".join(
            f"{n}: hip {h:.2e} (cap {grad_caps[_family(n)]:.1e}) ref32 {r:.2e}" for _, h, r, g, n in over[:10])
        worst_gross = max(report, key=lambda r: r[3] / cap_of(r[4], r[2]))
        assert worst_gross[3] <= 10 * cap_of(worst_gross[4], worst_gross[2]), \
            f"gross gradient mismatch: {worst_gross[4]} max-abs {worst_gross[3]:.2e}"
        return
    report.sort(reverse=True)
    summary = "; ".join(f"{n}: hip {h:.2e} ref32 {r:.2e} max {g:.2e}" for _, h, r, g, n in report[:10])
    assert max(r[3] for r in report) <= min(0.1 * grad_slack, 0.5), "gross gradient mismatch: " + summary
    by_err = sorted(report, key=lambda r: -r[1])[:6]
    summary += " || worst abs: " + "; ".join(f"{n}: hip {h:.2e} ref32 {r:.2e}" for _, h, r, g, n in by_err)
    over_hard = [r for r in report if r[1] > min(max(HARD_CAP, 3 * r[2] + GRAD_TOL) * grad_slack, 0.2)]
    assert not over_hard, "gradient rel-L2 error above the hard cap: " + "; ".join(
        f"{n}: hip {h:.2e} ref32 {r:.2e}" for _, h, r, g, n in over_hard[:8])
    # a ReLU-kink flip can land on either side (HIP or fp32 reference) and then dominates the handful
    # of tensors of that layer (and of everything downstream of it), so the calibrated bound is required
    # of >= (1 - OUTSIDE_FRACTION) of the tensors, not of all; the hard caps above still apply to every tensor
    outside = [r for r in report if r[0] > 1.0]
    assert len(outside) <= OUTSIDE_FRACTION * len(report), \
        f"{len(outside)}/{len(report)} gradients further from exact than the fp32 reference allows: " + summary
    med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
    assert med([r[1] for r in report]) <= 2 * med([r[2] for r in report]) + 1e-4 * grad_slack ** 2, \
        "median gradient error: " + summary


def test_emu_tiny_clip_forward_backward():
    """T=2, 64x64 frames, 3 text tokens through the host emulator."""
    dev = use_emu()
    torch.manual_seed(0)
    _compare(_run_hip(dev, 2, 64, 3, blocks=SMALL_NET), Ref.oracle(2, 64, 3, blocks=SMALL_NET))


def _train_step(dev, seed, T=2, res=64, L=3, p_override=None, backward=True):
    """one train-mode (dropout active) forward + loss + backward of the tiny clip"""
    from stcat_amd import ops
    text = synth.synth_text(L)
    model, criterion, wd = build_model(None, SyntheticText(text))
    synth.fill_module_(model)
    model.to(dev).train()
    if p_override is not None:
        for m in model.modules():
            if hasattr(m, "dropout_p"):
                m.dropout_p = p_override
    ops.manual_seed(seed)
    frames = synth.synth_frames(T, res).to(dev)
    mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
    out = model(NestedTensor(frames, mask, [T]), ["synthetic"])
    keep = {k: out[k].detach().cpu().clone() for k in ("pred_boxes", "pred_sted", "weights")}  # the loss edits out
    act, tb = synth.synth_targets(T)
    losses = criterion(out, [{"actioness": act.to(dev), "boxs": BoxList(tb, (res, res)).to(dev)}], [T])
    total = sum(losses[k] * wd[k] for k in losses)
    if not backward:       # (the runs that only compare outputs / the loss: forward is enough — the emulator is slow)
        return keep, total.item(), {}
    total.backward()
    grads = {n: q.grad.detach().cpu().clone() for n, q in model.named_parameters() if q.grad is not None}
    return keep, total.item(), grads


def _check_train_mode(dev):
    o1, l1, g1 = _train_step(dev, seed=11)
    o2, l2, g2 = _train_step(dev, seed=11)
    o3, l3, _ = _train_step(dev, seed=12, backward=False)
    # same seed -> the same masks in forward AND backward.  Not bitwise on the GPU: the split-K launches (weight
    # gradients, the skinny FFN forward) add partial sums atomically in arrival order — compare to fp32 round-off;
    # a different mask would move the outputs by O(1).
    for k in o1:
        close(o2[k], o1[k], 1e-5, "replayed output " + k)
        assert torch.equal(o1[k] == 0, o2[k] == 0), k      # the dropped positions themselves are identical
    assert abs(l1 - l2) <= 1e-5 * max(1.0, abs(l1))
    for n in g1:
        close(g2[n], g1[n], 1e-5, "replayed gradient " + n)
    assert all(torch.isfinite(v).all() for v in g1.values()) and math.isfinite(l1)
    assert not torch.equal(o1["pred_boxes"], o3["pred_boxes"]) and l1 != l3     # another seed, other masks
    # dropout really acts on the heads' outputs (net_utils.py:24: p=0.3 after the LAST layer too): exact zeros appear
    assert (o1["pred_sted"] == 0).any()
    # train mode with p = 0 is the eval arithmetic
    o0, l0, _ = _train_step(dev, seed=11, p_override=0.0, backward=False)
    model_eval = _run_hip_impl(dev, 2, 64, 3, with_backward="loss")
    close(o0["pred_boxes"], model_eval[0]["pred_boxes"], 1e-6, "p=0 train vs eval boxes")
    assert abs(l0 - model_eval[1]["total"]) <= 1e-5 * max(1.0, abs(l0))


class _HipMasks:
    """The dropout masks ONE HIP train-mode forward drew, handed to the CPU oracle site by site (oracle.dropout_sites).

    The kernels' decisions are a pure function of (seed, device base + host offset + element index) — csrc/stcat_rng.h,
    host twin ops.dropout_keep_mask — and ops.dropout_trace() lists (offset, decisions) of every site in launch order.
    The HIP path runs encoder -> time decoder (forked stream, issued first) -> box decoder -> heads; the reference runs
    the box decoder before the time decoder, so the trace is cut into sections by their site counts and every section is
    consumed in order; each hand-over checks the site's size, so a site added on one side only cannot go unnoticed.
    Layouts: HIP activations are batch-first rows [frames, tokens] where the reference is token-first [tokens, frames];
    self-attention decisions are stored [batch, head, key (padded to 32), query (padded)], the one-query
    cross-attention's [batch, head, key]."""

    SITES_PER_LAYER = {"enc": 4, "time": 6, "box": 6}      # probs, dropout1, (q1 probs, dropout3,) FFN inner, dropout2/4

    def __init__(self, trace, seed, base, n_enc=12, n_dec=6, n_heads=4):
        from collections import deque
        from stcat_amd import ops
        self.ops, self.seed, self.base = ops, seed, base
        cuts = [("enc", n_enc * 4), ("time", n_dec * 6), ("box", n_dec * 6), ("heads", n_heads)]
        assert len(trace) == sum(c for _, c in cuts), (len(trace), cuts)
        self.q, k = {}, 0
        for name, c in cuts:
            self.q[name] = deque(trace[k:k + c])
            k += c

    def _mask(self, section, n, p, dtype):
        off, numel = self.q[section].popleft()
        assert numel == n, f"dropout site of section {section}: the HIP path drew {numel} decisions, the oracle site has {n}"
        keep = self.ops.dropout_keep_mask(self.seed, self.base + off, n, p)
        scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))           # the kernels' fp32 1 / (1 - p)
        return torch.from_numpy(keep.astype(np.float64) * scale).to(dtype)

    def elementwise(self, section, x, p):
        m = self._mask(section, x.numel(), p, x.dtype)
        if x.dim() == 3:
            A, B, D = x.shape
            m = m.view(B, A, D).transpose(0, 1)
        else:
            m = m.view(x.shape)
        return x * m

    def probs(self, section, kind, pr, p, N, nh):
        NH, Lq, S = pr.shape
        assert NH == N * nh
        if kind == "self":
            assert Lq == S
            SP = (S + 31) // 32 * 32
            m = self._mask(section, N * nh * SP * SP, p, pr.dtype).view(N, nh, SP, SP)[:, :, :S, :S].transpose(-1, -2)
        else:
            assert Lq == 1
            m = self._mask(section, N * nh * S, p, pr.dtype).view(N, nh, 1, S)
        return pr * m.reshape(NH, Lq, S)

    def assert_all_consumed(self):
        left = {k: len(v) for k, v in self.q.items() if v}
        assert not left, f"HIP dropout sites the oracle never reached: {left}"


def _check_train_mode_against_oracle(dev, T, res, L, mma="f32", grad_caps=None):
    """Train mode at MODEL level against the oracle: one forward + loss + backward with dropout active
    (0.1 in 48 + 72 layer sites, 0.3 in the span / actioness heads), then the oracle run in fp32 and fp64 with the very
    masks the kernels drew.  Same bars as eval mode: outputs absolute 1e-3, span bit-exact, 30 loss terms, calibrated
    gradients."""
    from stcat_amd import _lib, ops
    _lib.set_mma_mode(mma)
    try:
        model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
        synth.fill_module_(model)
        model.to(dev).train()
        ops.manual_seed(1234)
        trace = ops.dropout_trace(True)
        frames, mask, H, W = _clip_of(T, res)
        out = model(NestedTensor(frames.to(dev), mask.to(dev), [T]), ["synthetic"])
        ops.dropout_trace(False)
        seed = ops.dropout_stream_state()[0]
        base = int(ops._dropout_stream.base(dev).item())
        keys = ("pred_boxes", "pred_sted", "pred_actioness", "weights")
        keep = {k: out[k].detach().cpu().clone() for k in keys}
        keep["aux"] = [{k: a[k].detach().cpu().clone() for k in keys} for a in out["aux_outputs"]]
        sizes = torch.tensor([[float(H), float(W)]], device=dev).repeat(T, 1)
        boxes, sted = build_postprocessors()(out, sizes, [list(range(100, 100 + T))], [T])
        keep["post_boxes"], keep["post_sted"] = boxes.cpu(), sted
        act, tb = synth.synth_targets(T)
        losses = criterion(out, [{"actioness": act.to(dev), "boxs": BoxList(tb, (W, H)).to(dev)}], [T])
        total = sum(losses[k] * wd[k] for k in losses)
        total.backward()
        grads = {n: q.grad.detach().cpu() for n, q in model.named_parameters() if q.grad is not None}
        losses = {k: v.item() for k, v in losses.items()}
        losses["total"] = total.item()
    finally:
        ops.dropout_trace(False)
        _lib.set_mma_mode("f32")
    assert len(trace) == 12 * 4 + 6 * 6 + 6 * 6 + 4
    ref = Ref.oracle(T, res, L, sites=lambda: _HipMasks(trace, seed, base))
    # the masks matter: the eval-mode oracle is far away from the train-mode outputs
    far = Ref.oracle(T, res, L, with_backward=False)
    assert (far.out["pred_sted"] - ref.out["pred_sted"]).abs().max() > 1e-2
    # grad_slack 2: with dropout a C1 gradient is a sum over ~10 % fewer samples and the run-to-run spread of the HIP
    # path itself (atomically ordered split-K sums -> single ReLU-kink flips in layer4's 7 x 7 maps) reaches 1.4e-1 of
    # a tensor's max-abs on one box in three (profiles/r04_gpu_tests.log); a wrong mask moves the OUTPUTS by O(1) and the
    # output / loss bars above are not relaxed
    _compare((keep, losses, grads), ref, grad_caps=grad_caps, grad_slack=2.0)


def test_emu_train_mode_against_oracle():
    with _depth(SMALL_NET):
        _check_train_mode_against_oracle(use_emu(), 2, 64, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("mma", [BENCH_MMA, "f16x3p"])
def test_gpu_train_mode_against_oracle(mma):
    """C1 (T=8, 224 x 224, L=10), dropout on, against the oracle fed with the kernels' masks — in the default arithmetic
    and in the experimental fp16-plane mode"""
    T, res, L = synth.CONFIGS["C1"]
    _check_train_mode_against_oracle(use_hip(), T, res, L, mma=mma)


def test_emu_train_mode_dropout():
    with _depth(SMALL_NET):
        _check_train_mode(use_emu())


def _check_two_forwards_before_backward(dev):
    """every train-mode forward opens a new dropout counter range; the backward of an EARLIER forward must
    still regenerate that forward's masks (gradient accumulation over two clips, loss A + loss B)."""
    from stcat_amd import ops
    T, res, L = 2, 64, 3
    model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
    synth.fill_module_(model)
    model.to(dev).train()
    frames = synth.synth_frames(T, res).to(dev)
    mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
    act, tb = synth.synth_targets(T)
    tgt = [{"actioness": act.to(dev), "boxs": BoxList(tb, (res, res)).to(dev)}]

    def fwd():
        out = model(NestedTensor(frames, mask, [T]), ["synthetic"])
        losses = criterion(out, tgt, [T])
        return sum(losses[k] * wd[k] for k in losses)

    def grads_of(total):
        for q in model.parameters():
            q.grad = None
        total.backward()
        return {n: q.grad.detach().cpu().clone() for n, q in model.named_parameters() if q.grad is not None}

    ops.manual_seed(21)
    g_alone = grads_of(fwd())                 # forward A, backward A
    ops.manual_seed(21)
    t_a = fwd()                               # forward A (same masks as above) ...
    t_b = fwd()                               # ... forward B draws new ones before A's backward runs
    g_first = grads_of(t_a)
    assert t_a.item() != t_b.item()
    for n in g_alone:
        close(g_first[n], g_alone[n], 1e-5, "gradient of the first of two forwards: " + n)
    del t_b


def test_emu_two_forwards_before_backward():
    with _depth(SMALL_NET):
        _check_two_forwards_before_backward(use_emu())


@pytest.mark.gpu
def test_gpu_two_forwards_before_backward():
    _check_two_forwards_before_backward(use_hip())


@pytest.mark.gpu
def test_gpu_train_mode_dropout():
    _check_train_mode(use_hip())


@pytest.mark.gpu
def test_gpu_c1_forward_backward(golden_dir):
    dev = use_hip()
    hip = _hip_case(dev, "C1")
    _compare(hip, Ref.fixture("C1"))
    # and against the round-1 fixture of the same run (stage tensors, all 626 gradient norms, eleven strided gradients)
    g = np.load(os.path.join(golden_dir, "C1.npz"))
    keep, losses, grads = hip
    for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights"):
        close(keep[k], torch.from_numpy(g[f"out/{k}"]), OUT_TOL, "golden " + k, absolute=True)
    assert keep["post_sted"] == g["post/sted"].tolist()
    close(keep["post_boxes"], torch.from_numpy(g["post/boxes"]), OUT_TOL, "golden post boxes")
    for k, v in zip(g["loss/keys"], g["loss/values"]):
        assert abs(losses[str(k)] - float(v)) <= 1e-3 * max(1.0, abs(float(v))), k
    norms = dict(zip([str(n) for n in g["grad/names"]], g["grad/norms"]))
    for n_, gr in grads.items():
        assert abs(gr.norm().item() - float(norms[n_])) <= 1e-2 * max(1.0, float(norms[n_])), n_
    # the eleven gradients the fixture holds in full (one per family, layer2.0 .. the heads): tensor against tensor.
    # The bound is what separates two fp32 evaluations of this chain at C1 (the fp32 CPU reference itself sits 1.2e-2 /
    # 2.5e-2 from an fp64 run on layer2 / layer3 tensors: single ReLU-kink flips among 8 x 28 x 28 samples): 3e-2 rel-L2
    # for the backbone, 5e-3 elsewhere; structurally zero gradients (key-side biases) get the absolute floor.
    full = [k[len("grad/full/"):] for k in g.files if k.startswith("grad/full/")]
    assert len(full) >= 11
    for n_ in full:
        ref = torch.from_numpy(g["grad/full/" + n_]).double()
        hip_name = "ground_decoder.decoder." + n_ if n_.startswith("bbox_embed.") and n_ not in grads else n_
        got = grads[hip_name].double().reshape(-1)
        if got.numel() > ref.numel():            # the fixture keeps a flat-stride sub-sample of large tensors
            got = got[::-(-got.numel() // (1 << 16))]
        assert got.shape == ref.shape, n_
        err = (got - ref).norm().item() / (ref.norm().item() + GRAD_ABS_FLOOR * ref.numel() ** 0.5 / GRAD_TOL)
        assert err <= (3e-2 if n_.startswith("vis_encoder.") else 5e-3), f"golden gradient {n_}: rel-L2 {err:.2e}"


@pytest.mark.gpu
@pytest.mark.parametrize("mma", ["bf16x3", "bf16x6", "bf16x3p", "bf16x6p", "f16x3p"])
def test_gpu_c1_split_bf16_modes(mma):
    """The split-bf16 GEMM modes must meet the same bars as the fp32-MFMA mode."""
    dev = use_hip()
    _compare(_hip_case(dev, "C1", mma=mma), Ref.fixture("C1"),
             grad_caps=GRAD_CAPS_16BIT if mma in ("bf16x3", "bf16x3p") else None)


def test_emu_tiny_clip_bf16x3_planes():
    """mma mode bf16x3p: the plane-format backbone (LDS-DMA staged GEMMs, transposing-read weight gradient) end to end."""
    dev = use_emu()
    # (tiny clip = wiring check: layer3 is a 4x4 map of 2 frames, one flipped ReLU kink moves a conv gradient by
    # several 1e-2; the residual stream also carries 16 instead of 24 significand bits here)
    _compare(_run_hip(dev, 2, 64, 3, mma="bf16x3p", blocks=SMALL_NET), Ref.oracle(2, 64, 3, blocks=SMALL_NET),
             grad_caps={k: min(16 * v, 0.1) for k, v in GRAD_CAPS_16BIT.items()})


def test_emu_tiny_clip_bf16x6_planes():
    """mma mode bf16x6p (the bench default): three-plane backbone (fp32 values exactly, six-term products), bf16x6 Linear
    layers, fp32-pipe attention — held to the calibrated fp32-class gradient bound, like f32 / bf16x6."""
    dev = use_emu()
    _compare(_run_hip(dev, 2, 64, 3, mma="bf16x6p", blocks=SMALL_NET), Ref.oracle(2, 64, 3, blocks=SMALL_NET))


@pytest.mark.skipif(not os.environ.get("STCAT_SLOW"), reason="all 33 bottlenecks through the emulator: ~3 minutes on 8 cores; "
                    "STCAT_SLOW=1 runs it (depth is covered by every GPU model test)")
def test_emu_tiny_clip_bf16x6_planes_full_depth():
    dev = use_emu()
    _compare(_run_hip(dev, 2, 64, 3, mma="bf16x6p"), Ref.oracle(2, 64, 3))


def test_emu_tiny_clip_f16x3_planes():
    """mma mode f16x3p (round 4, experimental): two fp16 planes per backbone tensor (22 significand bits, three products),
    weight / gradient planes scaled by powers of two into fp16's range — held to the calibrated fp32-class gradient bound"""
    dev = use_emu()
    _compare(_run_hip(dev, 2, 64, 3, mma="f16x3p", blocks=SMALL_NET), Ref.oracle(2, 64, 3, blocks=SMALL_NET))


def test_emu_tiny_clip_bf16x3():
    dev = use_emu()
    # T=2 frames of 64x64 (2x2 feature map): a tensor's gradient is a sum over a handful of tokens, so ONE flipped
    # ReLU kink moves it by ~1e-2 — the tiny clip checks wiring, the calibrated caps apply at C1 / C3 on the GPU
    _compare(_run_hip(dev, 2, 64, 3, mma="bf16x3", blocks=SMALL_NET), Ref.oracle(2, 64, 3, blocks=SMALL_NET),
             grad_caps={k: min(8 * v, 5e-2) for k, v in GRAD_CAPS_16BIT.items()})


@pytest.mark.gpu
def test_gpu_c2_forward():
    """BASELINE configs[1]: HC-STVG-like T=32, 416x416 (13x13 map, S=180), forward only, in the default arithmetic
    (bf16x6p) against the reference's own outputs (tests/golden/model_C2.npz)."""
    dev = use_hip()
    _compare(_hip_case(dev, "C2", mma=BENCH_MMA), Ref.fixture("C2"), with_backward=False)


def _print_gradient_report(tag, rows):
    """distribution of the per-tensor gradient errors of a run (shows up in the kept test log with `pytest -s` / -rP)"""
    if not rows:
        return
    e = sorted(r[1] for r in rows)
    outside = sum(1 for r in rows if r[0] > 1.0)
    worst = max(rows, key=lambda r: r[1])
    gross = max(r[3] for r in rows)
    strict = sum(1 for r in rows if r[1] > 3 * r[2] + GRAD_TOL)
    print(f"[gradient report] {tag}: {len(rows)} tensors, rel-L2 vs fp64 median {e[len(e) // 2]:.2e}, p90 "
          f"{e[int(0.9 * len(e))]:.2e}, max {e[-1]:.2e} ({worst[4]}; fp32 reference there {worst[2]:.2e}); "
          f"{outside} outside the test's bound, {strict} outside 3 x e_ref + 1e-3; worst max-abs error / max-abs {gross:.2e}")


@pytest.mark.gpu
def test_gpu_c3_full_size_forward_backward():
    """The number bench.py sells is fwd + loss + BACKWARD at C3 in the fp32-class arithmetic bf16x6p: check exactly
    that, at full size, against the CPU oracle — outputs within an absolute 1e-3, span bit-exact, 30 loss terms, and the
    gradients held to the CALIBRATED fp32 bound (grad_caps=None: as close to the fp64 oracle run as the fp32 CPU
    reference itself is), not to per-family caps."""
    dev = use_hip()
    _compare(_hip_case(dev, "C3", mma=BENCH_MMA), Ref.fixture("C3"))


@pytest.mark.gpu
def test_gpu_c3_full_size_forward_backward_throughput_mode():
    """bench.py's `throughput_mode` (bf16x3p, 16 significand bits per operand) at full size: same output / span / loss
    bars, every gradient tensor within its measured family cap of the fp64 oracle run."""
    dev = use_hip()
    _compare(_hip_case(dev, "C3", mma=THROUGHPUT_MMA), Ref.fixture("C3"), grad_caps=GRAD_CAPS_16BIT)


def _run_bench_step(dev, name, mma, steps=3, train=False, use_plans=True, zero_dropout=False, pipeline=False, trace=None):
    """bench.py's OWN step object (stcat_amd/harness.py: TrainStep — bucketed reducer, zero arena, per-step loss plan)
    under launch plans, on the clip / targets of a fixture: step 1 runs eager, step 2 records, step 3 REPLAYS.  Returns
    the replayed step in the form of _run_hip (outputs before the criterion edits them, PostProcess, losses, gradients).
    pipeline: bench.py's default schedule (round 6) — every step declares the next step's frames and their frozen prefix
    runs under this step's grounding section; one more step, because the first one computes its prefix in place (eager,
    eager with a staged prefix, record, REPLAY).  trace: a dict that receives the dropout stream of the LAST step —
    seed, device base, and the (host offset, decisions) list of every site (taken on step 1: a replay draws the same
    offsets from the plan, the host-side take() does not run)."""
    from stcat_amd import _lib, ops, plans
    from stcat_amd.harness import TrainStep
    if pipeline:
        steps += 1
    T, res, L, pad, _ = synth.MODEL_CASES[name]
    frames, mask, H, W = _clip_of(T, res, pad)
    act, tb = synth.synth_targets(T)
    _lib.set_mma_mode(mma)
    plans.clear()
    plans.enable(use_plans)
    plans.STATS.update(recorded=0, replayed=0, eager=0, run_s=0.0, refused=0)
    ts = None
    try:
        ts = TrainStep(dev, (T, res, L), train=train, clip=(frames, mask),
                       targets=[{"actioness": act, "boxs": BoxList(tb, (W, H))}], pipeline_prefix=pipeline)
        if zero_dropout:           # train mode with every dropout probability at 0: the train-mode code path, eval-mode numbers
            for m in ts.model.modules():
                if hasattr(m, "dropout_p"):
                    m.dropout_p = 0.0
        ts.keep_outputs = True
        for k in range(steps):
            before = dict(plans.STATS)
            if trace is not None and k == 0:
                sites = ops.dropout_trace(True)
            total = ts.step()
            if trace is not None and k == 0:
                ops.dropout_trace(False)
                trace["sites"] = [(int(o), int(n)) for o, n in sites]
        if trace is not None:
            trace["seed"] = int(ops.dropout_stream_state()[0])
            trace["base"] = int(ops._dropout_stream.base(dev).item())
            trace["steps"] = steps
        if pipeline:
            st = ts.model.vis_encoder[0].prefix_stats
            assert st["inline"] == 1 and st["taken"] == steps - 1, st
        if use_plans:
            # the last step replayed every composite node, forward and backward: nothing ran eager, nothing was recorded
            assert plans.STATS["replayed"] - before["replayed"] >= 8, (before, plans.STATS)
            # (pipeline: the staged prefix is a launch plan per resident buffer — eager, eager, recorded, recorded over the
            #  four steps — so the LAST step records the second buffer's plan: a real execution, queued for the step after
            #  it; the composite nodes of the step under test are all replays)
            assert plans.STATS["recorded"] - before["recorded"] <= (1 if pipeline else 0), (before, plans.STATS)
            assert plans.STATS["eager"] == before["eager"], (before, plans.STATS)
            assert not plans.STATS.get("refused"), plans.STATS
        keep = {k: v.cpu() for k, v in ts.last_out.items() if torch.is_tensor(v)}
        keep["aux"] = [{k: v.cpu() for k, v in a.items()} for a in ts.last_out["aux"]]
        sizes = torch.tensor([[float(H), float(W)]], device=dev).repeat(T, 1)
        boxes, sted = build_postprocessors()({"pred_sted": ts.last_out["pred_sted"], "pred_boxes": ts.last_out["pred_boxes"]},
                                             sizes, [list(range(100, 100 + T))], [T])
        keep["post_boxes"], keep["post_sted"] = boxes.cpu(), sted
        losses = {k: v.item() for k, v in ts.last_losses.items()}
        losses["total"] = total.item()
        grads = {n: g.cpu() for n, g in ts.gradients().items()}
        return keep, losses, grads
    finally:
        plans.enable(False)
        if ts is not None:
            ts.close()
        _lib.set_mma_mode("f32")


@pytest.mark.gpu
def test_gpu_c3_replayed_bench_step_fp16_planes():
    """the same replayed C3 step in the experimental mode f16x3p (what `near_f32_mode` of the bench line times)"""
    dev = use_hip()
    _compare(_run_bench_step(dev, "C3", "f16x3p"), Ref.fixture("C3"))


@pytest.mark.gpu
def test_gpu_c3_replayed_bench_step():
    """the path bench.py TIMES — its own step object, launch plans on, the third step replayed from
    the recorded C++ launch sequences out of the private memory pool, three streams, the reducer's buckets — at the
    benchmark size C3 in the default arithmetic, against the REFERENCE's fixture: outputs absolute 1e-3, span bit-exact,
    30 loss terms, every gradient tensor under the calibrated fp32 bound.  (Eval mode: the reference's dropout stream
    cannot be reproduced; the train-mode masks are checked against the oracle in test_gpu_train_mode_against_oracle.)"""
    dev = use_hip()
    _compare(_run_bench_step(dev, "C3", BENCH_MMA), Ref.fixture("C3"))


@pytest.mark.gpu
def test_gpu_c3_train_mode_bench_step_plans_equal_eager():
    """what bench.py TIMES is the train-mode step (dropout 0.1 / 0.3 on); the fixture comparison above is
    eval mode and the mask-fed oracle comparison runs at C1.  At the benchmark size the train-mode step is held to itself:
    the same seed, three steps each, once replayed from the launch plans and once eager — the third steps must agree
    (outputs, 30 losses, every gradient; 1e-5 of tensor scale: the split-K weight gradients sum atomically)."""
    dev = use_hip()
    a = _run_bench_step(dev, "C3", BENCH_MMA, train=True, use_plans=True)
    b = _run_bench_step(dev, "C3", BENCH_MMA, train=True, use_plans=False)
    for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights"):
        sc = b[0][k].abs().max().item() + 1e-12
        assert (a[0][k] - b[0][k]).abs().max().item() <= 1e-5 * max(sc, 1.0), k
    assert a[0]["post_sted"] == b[0]["post_sted"]
    for k, v in b[1].items():
        assert abs(a[1][k] - v) <= 1e-5 * max(1.0, abs(v)), (k, a[1][k], v)
    assert a[2].keys() == b[2].keys()
    worst, where = 0.0, None
    for n, g in b[2].items():
        sc = g.abs().max().item()
        # (1e-7 absolute: the key-side biases of every attention have an analytically ZERO gradient — softmax is invariant to
        #  a shift of all scores of a row — and hold 1e-9 of summation noise that differs between any two runs)
        ratio = (a[2][n] - g).abs().max().item() / (sc + 1e-7 / 1e-3)
        if ratio > worst:
            worst, where = ratio, n
    print(f"[train-mode C3] replayed vs eager: worst gradient difference {worst:.2e} of the tensor's scale ({where})")
    # (bar: two EAGER runs of one step differ by up to ~1e-3 on single encoder-FFN tensors — atomically ordered split-K sums
    #  move a pre-activation across its ReLU kink here and there; seen in this test: 1.1e-5 .. 2.2e-4)
    assert worst <= 1e-3, (worst, where)


def _check_train_mode_bench_step_against_fixture(dev, case):
    """the mode bench.py times (train: dropout 0.1 / 0.3 on) at the size it times, through the path it times
    (its step object, pipelined prefix, launch plans, the REPLAYED step) against expected values from the oracle: the
    fixture holds the dropout stream it was computed with (seed, device base, every site's offset and size); the live
    step must draw EXACTLY that stream — otherwise the comparison would be against other masks — and is then held to the
    usual bars: outputs 1e-3 absolute, span bit-exact, 30 losses, calibrated gradients with grad_slack = 1."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"model_{case}_train.npz")
    g = np.load(path)
    tr = {}
    got = _run_bench_step(dev, case + "_train", BENCH_MMA, train=True, pipeline=True, trace=tr)
    assert tr["seed"] == int(g["dropout/seed"]) and tr["base"] == int(g["dropout/base"]), (tr["seed"], tr["base"])
    assert [list(x) for x in tr["sites"]] == g["dropout/sites"].tolist(), "the live dropout stream differs from the fixture's"
    rows = []
    _compare(got, Ref.fixture(case + "_train"), report_to=rows)
    return rows


@pytest.mark.gpu
def test_gpu_c3_train_mode_bench_step_against_fixture():
    _check_train_mode_bench_step_against_fixture(use_hip(), "C3")


@pytest.mark.gpu
def test_gpu_c1_train_mode_bench_step_against_fixture():
    _check_train_mode_bench_step_against_fixture(use_hip(), "C1")


@pytest.mark.gpu
def test_gpu_c3_train_mode_zero_dropout_equals_reference_fixture():
    """... and the train-mode code path itself (module.train(), every site's dropout probability set to 0: the fused
    dropout epilogues, LayerNorm+dropout, attention-probability dropout all run with p = 0) reproduces the reference's
    eval-mode fixture at the benchmark size, replayed from the plans"""
    dev = use_hip()
    _compare(_run_bench_step(dev, "C3", BENCH_MMA, train=True, zero_dropout=True), Ref.fixture("C3"))


@pytest.mark.gpu
def test_gpu_c1_replayed_bench_step():
    """the same at C1 (T=8, 224 x 224), where a step is almost pure launch sequence"""
    dev = use_hip()
    _compare(_run_bench_step(dev, "C1", BENCH_MMA), Ref.fixture("C1"))


@pytest.mark.gpu
def test_gpu_c3_full_size_forward_backward_fp16_planes():
    """mode f16x3p (round 4, experimental: two fp16 planes = 22 significand bits, three products, power-of-two operand
    scales) at the benchmark size against the reference fixture, held to the CALIBRATED fp32-class gradient bound — the
    bound `bf16x6p` and the exact-fp32 mode are held to, not the 16-bit modes' family caps"""
    dev = use_hip()
    _compare(_hip_case(dev, "C3", mma="f16x3p"), Ref.fixture("C3"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["C2", "C5", "SQ8_ragged", "NS8", "NS8_ragged"])
def test_gpu_fp16_planes_other_cases(case):
    """mode f16x3p on the other reference fixtures: HC-STVG-shaped forward (C2), the long clip (C5: T=128, L=40), padded and
    non-square clips with backward — the ranges that decide whether fp16 planes overflow or go subnormal differ from clip to
    clip, the scales are constants: same bars as the default arithmetic, calibrated gradient bound"""
    dev = use_hip()
    bwd = synth.MODEL_CASES[case][4]
    _compare(_hip_case(dev, case, mma="f16x3p"), Ref.fixture(case), with_backward=bwd)


# ---- non-square and padded clips ---------------------------------------------------------------------
# The reference's transforms (datasets/build.py:20-44: short side 448, max_size 720) turn a 16:9 video into 405 x 720
# frames: a 13 x 23 layer4 map, 13*23 + L + 1 = 310 tokens per frame in the encoder, 309 keys per frame in the decoders'
# time-aligned cross-attention — more than the 256 the round-2 kernels were built for.  Padding masks
# (modal_encoder.py:46, query_decoder.py:111) are exercised at model level here, not only per op.
def test_emu_nonsquare_padded_clip_forward_backward():
    """96 x 160 frames (3 x 5 map) with a ragged padding mask, through the host emulator: H != W everywhere, key-padding
    in every attention of the assembled model, forward + loss + backward against the oracle."""
    dev = use_emu()
    _compare(_run_hip(dev, 3, (96, 160), 3, pad="ragged", blocks=SMALL_NET),
             Ref.oracle(3, (96, 160), 3, pad="ragged", blocks=SMALL_NET))


@pytest.mark.gpu
def test_gpu_nonsquare_405x720_padded_clip_forward_backward():
    """T=8 frames of 405 x 720 (what the reference's resize makes of a 16:9 video), L=10, ragged padding on three frames:
    310 encoder tokens / 309 decoder keys per frame -> the long-row self-attention kernels and the two-chunk one-query
    cross-attention, odd H (405 -> 203 -> 102 -> 51 -> 26 -> 13) and W % 32 != 0 in every conv.  Forward + loss +
    backward in the bench arithmetic vs the CPU oracle: outputs 1e-3 absolute, span bit-exact, calibrated gradients."""
    dev = use_hip()
    # (round 3 compared with the oracle and needed grad_slack 4 here; against the reference's fixture the padded clip meets
    # the plain calibrated bound: worst tensor 1.5e-2 where the fp32 reference is 3.0e-3 from its fp64 run, 1 of 626 outside)
    _compare(_hip_case(dev, "NS8_ragged", mma=BENCH_MMA), Ref.fixture("NS8_ragged"))


@pytest.mark.gpu
def test_gpu_nonsquare_405x720_clip_forward_backward():
    """the same 405 x 720 clip without padding (mask all False, what the reference's loader produces with one video per
    rank): isolates the non-square geometry from the padding"""
    dev = use_hip()
    _compare(_hip_case(dev, "NS8", mma=BENCH_MMA), Ref.fixture("NS8"))


@pytest.mark.gpu
def test_gpu_square_clip_with_padding_mask():
    """C1-sized square clip whose mask is NOT all-zero (three partially padded frames): fwd + loss + bwd, bench arithmetic"""
    dev = use_hip()
    _compare(_hip_case(dev, "SQ8_ragged", mma=BENCH_MMA), Ref.fixture("SQ8_ragged"))


@pytest.mark.gpu
def test_gpu_nonsquare_clip_throughput_mode():
    """the same 405 x 720 clip in the 16-bit throughput mode: rows longer than 256 tokens train through the fp32 long-row
    attention kernels there too (ops.MhaSelfFn); forward + loss + backward, family caps on the gradients"""
    dev = use_hip()
    _compare(_hip_case(dev, "NS8_ragged", mma=THROUGHPUT_MMA), Ref.fixture("NS8_ragged"),
             grad_caps={k: min(4 * v, 5e-2) for k, v in GRAD_CAPS_16BIT.items()})  # (T = 8: few samples per gradient)


@pytest.mark.gpu
def test_gpu_c5_full_size_forward():
    """BASELINE configs[4] at FULL size: T=128, 448x448, 40 text tokens (S = 237 tokens per frame, 8 key tiles),
    forward + PostProcess in the bench arithmetic against the CPU oracle."""
    dev = use_hip()
    ref = Ref.fixture("C5")
    _compare(_hip_case(dev, "C5", mma=BENCH_MMA), ref, with_backward=False)
    # ... and as BASELINE configs[4] words it: with the decoder replayed from its hipGraph
    _compare(_hip_case(dev, "C5", mma=BENCH_MMA, graphed=True), ref, with_backward=False)


@pytest.mark.gpu
def test_gpu_captured_decoder_equals_eager():
    """STCATNet.capture_decoder: the hipGraph replay of decoder + heads (fed through static input buffers, captured on a
    different clip) gives the eager launches' outputs — square clip, and a padded non-square one (key-chunked one-query
    attention, padding masks inside the captured region)"""
    dev = use_hip()
    for T, res, pad in ((8, 224, None), (4, (405, 720), "ragged")):
        eager = _run_hip(dev, T, res, 10, with_backward=False, mma=BENCH_MMA, pad=pad)[0]
        rep = _run_hip(dev, T, res, 10, with_backward=False, mma=BENCH_MMA, pad=pad, graphed=True)[0]
        for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights", "post_boxes"):
            # (not bitwise: the eager run goes through the training-form nodes, and split-K sums are added in arrival order)
            close(rep[k], eager[k], 1e-5, f"captured decoder {k} (T={T}, {res})")
        assert rep["post_sted"] == eager["post_sted"]
        for a, b in zip(rep["aux"], eager["aux"]):
            for k in a:
                close(a[k], b[k], 1e-5, f"captured decoder aux {k}")


@pytest.mark.gpu
def test_gpu_uint8_frames_through_the_model():
    """the model fed with the decoder's uint8 [T,H,W,3] frames (ToTensor + Normalize fused into the
    stem gather, datasets/transforms.py:155-168) gives the outputs of the fp32 path on the normalised [T,3,H,W] tensor."""
    from stcat_amd import _lib, ops
    dev = use_hip()
    T, res, L = 8, 224, 10
    _lib.set_mma_mode(BENCH_MMA)
    try:
        model, _, _ = build_model(None, SyntheticText(synth.synth_text(L)))
        model.eval()
        synth.fill_module_(model)
        model.to(dev)
        u8 = torch.randint(0, 256, (T, res, res, 3), generator=torch.Generator().manual_seed(5), dtype=torch.uint8)
        mean, std = torch.tensor(ops.PIXEL_MEAN), torch.tensor(ops.PIXEL_STD)
        f32 = ((u8.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()
        mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
        with torch.no_grad():
            o8 = model(NestedTensor(u8.to(dev), mask, [T]), ["q"])
            of = model(NestedTensor(f32.to(dev), mask, [T]), ["q"])
    finally:
        _lib.set_mma_mode("f32")
    # The two stems round differently ((u/255 - mean)/std inside the gather vs. the pre-normalised tensor): 3e-3 of 117
    # on the layer4 features in the bench's 16-bit-significand arithmetic, which reaches the heads as 2e-5 (boxes) to
    # 4e-4 (actioness logits, scale 4.8); in exact-fp32 mode the same comparison gives 1e-6 .. 1e-5 (tools/u8_check.py).
    # The bar is the north star's absolute 1e-3.
    for k in ("pred_boxes", "pred_sted", "pred_actioness"):
        close(o8[k], of[k], 1e-3, "uint8 input " + k, absolute=True)


@pytest.mark.gpu
def test_gpu_two_pass_eval_path():
    """engine/evaluate.py:97-119: even/odd frame halves evaluated separately, spans united."""
    from stcat_amd.pipeline import evaluate_video
    from stcat_amd import _lib
    dev = use_hip()
    T, res, L = 8, 224, 10
    _lib.set_mma_mode(BENCH_MMA)
    try:
        model, _, _ = build_model(None, SyntheticText(synth.synth_text(L)))
        model.eval()
        synth.fill_module_(model)
        model.to(dev)
        frames = synth.synth_frames(T, res).to(dev)
        mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
        sizes = torch.tensor([[float(res), float(res)]], device=dev).repeat(T, 1)
        ids = [list(range(100, 100 + T))]
        boxes, union = evaluate_video(model, build_postprocessors(), NestedTensor(frames, mask, [T]), ["q"], sizes, ids)
        # the same with gaps in the frame ids: engine/evaluate.py:114 interpolates the frames the sampler skipped
        gap_ids = [[100, 101, 103, 104, 108, 109, 110, 115]]
        dense, _ = evaluate_video(model, build_postprocessors(), NestedTensor(frames, mask, [T]), ["q"], sizes, gap_ids,
                                  interpolate=True)
    finally:
        _lib.set_mma_mode("f32")
    assert sorted(f for _, f in boxes) == ids[0] and all(b.shape == (4,) for b in boxes.values())
    sd = synth.synth_state_dict()
    spans = []
    for start in (0, 1):
        out = O.stcat_forward(sd, synth.synth_frames(T, res)[start::2], torch.zeros(T // 2, res, res, dtype=torch.bool),
                              synth.synth_text(L))
        b, sted, _ = O.post_process(out["pred_sted"], out["pred_boxes"], torch.ones(T // 2, 2) * res, ids[0][start::2],
                                    T // 2)
        spans.append(sted)
        for k, f in enumerate(ids[0][start::2]):
            close(boxes[(0, f)], b[k], OUT_TOL * res, f"eval box frame {f}")
    assert union == [[min(spans[0][0], spans[1][0]), max(spans[0][1], spans[1][1])]]
    # interpolated boxes == the oracle's linear_interp of the merged per-frame boxes
    assert sorted(f for _, f in dense) == list(range(100, 116))
    merged = {f: [[float(v) for v in boxes[(0, ids[0][k])].double().cpu()]] for k, f in enumerate(gap_ids[0])}
    want = O.linear_interp(merged)
    for f in range(100, 116):
        # (two separate runs of the model: split-K launches add partial sums in arrival order, so the given frames
        # themselves agree to fp32 round-off, not bitwise)
        close(dense[(0, f)], torch.tensor(want[f][0], dtype=torch.float64), 2e-5, f"interpolated box {f}")


def _trajectory(dev, mma, steps, T, res, L, lr):
    """`steps` training steps (train mode: dropout on, the counter-based masks are the same function of (seed, counter)
    in every arithmetic mode; clip 0.1 + AdamW through stcat_amd.optim, two alternating clips) -> the loss of every step"""
    from stcat_amd import _lib, ops, optim
    _lib.set_mma_mode(mma)
    try:
        model, criterion, wd = build_model(None, SyntheticText(synth.synth_text(L)))
        synth.fill_module_(model)
        model.to(dev).train()
        ops.manual_seed(5)
        opt = optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=lr, weight_decay=1e-4)
        clips = []
        for s in range(2):
            act, tb = synth.synth_targets(T, seed=s)
            clips.append((synth.synth_frames(T, res, seed=s).to(dev),
                          [{"actioness": act.to(dev), "boxs": BoxList(tb, (res, res)).to(dev)}]))
        mask = torch.zeros(T, res, res, dtype=torch.bool, device=dev)
        losses = []
        for s in range(steps):
            frames, tgt = clips[s % 2]
            opt.zero_grad()
            out = model(NestedTensor(frames, mask, [T]), ["synthetic"])
            l = criterion(out, tgt, [T])
            total = sum(l[k] * wd[k] for k in l)
            total.backward()
            opt.step(max_grad_norm=0.1)
            losses.append(total.item())
        return losses
    finally:
        _lib.set_mma_mode("f32")


@pytest.mark.gpu
def test_gpu_loss_trajectory_of_the_modes():
    """several optimizer steps, not one gradient.  Eight train-mode steps at C1 (T=8, 224 x 224) with
    the fused clip + AdamW tail in the exact-fp32 mode, in the fp32-class default `bf16x6p` and in the 16-bit throughput
    mode: the loss sequences stay together — fp32-class within 2e-3 relative at every step (run-to-run noise of the
    exact mode itself: atomically ordered split-K sums, ReLU-kink flips, amplified by the optimizer — measured 2.7e-4;
    bf16x6p 6.8e-4, bf16x3p 2.7e-3), the throughput mode within 1e-2 — and all of them go down."""
    dev = use_hip()
    T, res, L = synth.CONFIGS["C1"]
    steps, lr = 8, 2e-6
    ref = _trajectory(dev, "f32", steps, T, res, L, lr)
    again = _trajectory(dev, "f32", steps, T, res, L, lr)
    x6 = _trajectory(dev, BENCH_MMA, steps, T, res, L, lr)
    x3 = _trajectory(dev, THROUGHPUT_MMA, steps, T, res, L, lr)
    h3 = _trajectory(dev, "f16x3p", steps, T, res, L, lr)        # round 4: two fp16 planes, held to the fp32-class bar
    rel = lambda a, b: max(abs(p - q) / max(1.0, abs(q)) for p, q in zip(a, b))   # noqa: E731
    noise = rel(again, ref)
    print(f"loss trajectories: f32 {ref}\n  f32 again (noise {noise:.2e})\n  bf16x6p {x6} (dev {rel(x6, ref):.2e})\n"
          f"  bf16x3p {x3} (dev {rel(x3, ref):.2e})")
    assert all(map(lambda v: v == v and abs(v) < 1e6, ref + x6 + x3))
    assert rel(x6, ref) <= max(2e-3, 5 * noise), (x6, ref)
    print(f"  f16x3p {h3} (dev {rel(h3, ref):.2e})")
    assert all(map(lambda v: v == v and abs(v) < 1e6, h3)) and rel(h3, ref) <= max(2e-3, 5 * noise), (h3, ref)
    assert rel(x3, ref) <= max(1e-2, 10 * noise), (x3, ref)
    for tr in (ref, x6, x3):            # same clip, six optimizer steps later: the loss went down
        assert tr[6] < tr[0] and tr[7] < tr[1], tr
