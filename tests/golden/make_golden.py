#!/usr/bin/env python3
"""Generate golden vectors by importing the reference (runs ONLY in the build
container, where /root/reference exists; the GPU box never sees it).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What runs: the reference's own ``models.build_model(cfg)`` → ``STCATNet`` +
``VideoSTGLoss`` + ``PostProcess`` (models/__init__.py:5-41) on CPU, eval mode
(dropout off), filled with the deterministic synthetic weights
of ``stcat_amd.synth`` and fed the synthetic clip/text/targets of the same
module.  Stubs are injected only for packages that are absent from this image
and sit *outside* the hot path: yacs (config container), torchvision (the
ResNet-101 topology — third-party arithmetic, restated below from the
torchvision 0.11 definition; FrozenBatchNorm2d/BackboneBase/Joiner are the
reference's own), pytorch_pretrained_bert / torchtext / transformers (text
encoder, out of scope: replaced by a module that returns the synthetic text
boundary tensors).

The fixtures are data only: inputs are regenerated from names/seeds, outputs
are stored (sub-sampled where large).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)

from stcat_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------
# stubs for absent third-party packages
# ----------------------------------------------------------------------------
class CfgNode(dict):
    """Minimal yacs.config.CfgNode: attribute access + clone/merge/freeze."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return CfgNode({k: (v.clone() if isinstance(v, CfgNode) else v) for k, v in self.items()})

    def freeze(self):
        pass

    def defrost(self):
        pass

    def dump(self):
        return repr(self)

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self:
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                old = self.get(k)
                if isinstance(old, float) and isinstance(v, str):
                    v = float(v)  # yacs coerces '1e-5' style yaml strings
                self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v


class _Bottleneck(nn.Module):
    """torchvision 0.11 Bottleneck, v1.5 (stride on the 3x3 conv)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, norm_layer):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class _ResNet(nn.Module):
    def __init__(self, layers, norm_layer):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make(64, layers[0], 1, norm_layer)
        self.layer2 = self._make(128, layers[1], 2, norm_layer)
        self.layer3 = self._make(256, layers[2], 2, norm_layer)
        self.layer4 = self._make(512, layers[3], 2, norm_layer)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)

    def _make(self, planes, blocks, stride, norm_layer):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                               norm_layer(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, ds, norm_layer)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self.inplanes, planes, 1, None, norm_layer))
        return nn.Sequential(*layers)


def _resnet101(pretrained=False, replace_stride_with_dilation=None, norm_layer=None, **kw):
    assert not any(replace_stride_with_dilation or [False])
    return _ResNet([3, 4, 23, 3], norm_layer)


class _IntermediateLayerGetter(nn.ModuleDict):
    """torchvision.models._utils.IntermediateLayerGetter."""

    def __init__(self, model, return_layers):
        orig = dict(return_layers)
        layers = {}
        remaining = dict(return_layers)
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = {}
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    yc = mod("yacs.config", CfgNode=CfgNode)
    mod("yacs", config=yc)
    tv_models_utils = mod("torchvision.models._utils", IntermediateLayerGetter=_IntermediateLayerGetter)
    tv_models = mod("torchvision.models", resnet101=_resnet101, _utils=tv_models_utils)
    tv_boxes = mod("torchvision.ops.boxes",
                   box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    tv_ops = mod("torchvision.ops", boxes=tv_boxes)
    mod("torchvision", models=tv_models, ops=tv_ops)
    ppb_m = mod("pytorch_pretrained_bert.modeling", BertModel=object)
    mod("pytorch_pretrained_bert", modeling=ppb_m)
    mod("torchtext")
    mod("transformers", RobertaModel=object, RobertaTokenizerFast=object)


class SyntheticText(nn.Module):
    """Stands in for the out-of-scope text encoder: returns the synthetic
    boundary tensors (language_model/bert.py:59-74 output contract)."""

    def __init__(self, text):
        super().__init__()
        self.text = text

    def forward(self, texts, device):
        return self.text


def build_reference(L: int):
    install_stubs()
    sys.path.insert(0, REF)
    from config import cfg as _cfg  # noqa
    cfg = _cfg.clone()
    cfg.merge_from_file(os.path.join(REF, "experiments/VidSTG/e2e_STCAT_R101_VidSTG.yaml"))
    import models.pipeline as pipeline
    text = synth.synth_text(L)
    pipeline.build_text_encoder = lambda cfg: SyntheticText(text)
    from models import build_model, build_postprocessors
    model, criterion, weight_dict = build_model(cfg)
    model.eval()
    synth.fill_module_(model)
    return cfg, model, criterion, weight_dict, build_postprocessors(), text


def sub(t: torch.Tensor, max_elems: int = 1 << 16) -> np.ndarray:
    """Deterministic sub-sample of a large tensor (flat stride)."""
    f = t.detach().reshape(-1)
    if f.numel() <= max_elems:
        return f.numpy().copy()
    step = -(-f.numel() // max_elems)
    return f[::step].numpy().copy()


def run_config(name: str, with_backward: bool):
    T, res, L = synth.CONFIGS[name]
    cfg, model, criterion, weight_dict, post, text = build_reference(L)
    from utils.misc import NestedTensor
    from utils.bounding_box import BoxList

    frames = synth.synth_frames(T, res)
    mask = torch.zeros(T, res, res, dtype=torch.bool)
    videos = NestedTensor(frames, mask, [T])
    act, boxes = synth.synth_targets(T)
    targets = [{"actioness": act, "boxs": BoxList(boxes, (res, res), mode="xyxy")}]

    # stage boundaries via forward hooks on the reference modules
    stages = {}

    def hook(key):
        def f(mod, inp, out):
            stages[key] = out
        return f

    body = model.vis_encoder[0].body
    for ln in ("layer1", "layer2", "layer3", "layer4"):
        getattr(body, ln).register_forward_hook(hook(ln))
    model.vis_encoder.register_forward_hook(hook("vis_encoder"))
    model.input_proj.register_forward_hook(hook("input_proj"))
    model.ground_encoder.register_forward_hook(hook("ground_encoder"))
    model.ground_decoder.register_forward_hook(hook("ground_decoder"))
    model.ground_decoder.template_generator.register_forward_hook(hook("template"))

    ctx = torch.enable_grad() if with_backward else torch.no_grad()
    with ctx:
        out = model(videos, ["synthetic query"])
    g = {}
    for ln in ("layer1", "layer2", "layer3", "layer4"):
        g[f"stage/{ln}"] = sub(stages[ln])
        g[f"stage/{ln}/absmax"] = np.float32(stages[ln].abs().max().item())
    g["stage/vis_pos"] = sub(stages["vis_encoder"][1])
    g["stage/input_proj"] = sub(stages["input_proj"])
    mc = stages["ground_encoder"]
    g["stage/encoded_memory"] = sub(mc["encoded_memory"])
    g["stage/frames_cls"] = mc["frames_cls"].detach().numpy()
    g["stage/videos_cls"] = mc["videos_cls"].detach().numpy()
    g["stage/pos_query"] = stages["template"][0].detach().numpy()
    (hs, ref), (time_hs, weights) = stages["ground_decoder"]
    g["stage/hs"] = hs.detach().numpy()
    g["stage/ref"] = ref.detach().numpy()
    g["stage/time_hs"] = time_hs.detach().numpy()
    g["stage/weights"] = weights.detach().numpy()
    for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights"):
        g[f"out/{k}"] = out[k].detach().numpy().copy()
        for i, aux in enumerate(out["aux_outputs"]):
            g[f"out/aux{i}/{k}"] = aux[k].detach().numpy().copy()

    # post-process BEFORE the criterion (which slices pred_boxes in place, criterion.py:168-171)
    sizes = torch.tensor([[float(res), float(res)]]).repeat(T, 1)
    frame_ids = [list(range(100, 100 + T))]
    pb, steds = post(out, sizes, frame_ids, [T])
    g["post/boxes"] = pb.numpy()
    g["post/sted"] = np.asarray(steds, dtype=np.int64)

    if with_backward:
        losses = criterion(out, targets, [T])
        assert set(losses.keys()) == set(weight_dict.keys())
        total = sum(losses[k] * weight_dict[k] for k in losses)
        total.backward()
        keys = sorted(losses.keys())
        g["loss/keys"] = np.asarray(keys)
        g["loss/values"] = np.asarray([losses[k].item() for k in keys], dtype=np.float32)
        g["loss/weights"] = np.asarray([float(weight_dict[k]) for k in keys], dtype=np.float32)
        g["loss/total"] = np.float32(total.item())
        names, norms, no_grad = [], [], []
        for n_, p in model.named_parameters():
            if n_.startswith("text_encoder."):
                continue
            if p.grad is None:
                if p.requires_grad:
                    no_grad.append(n_)
                continue
            names.append(n_)
            norms.append(p.grad.norm().item())
        g["grad/names"] = np.asarray(names)
        g["grad/norms"] = np.asarray(norms, dtype=np.float32)
        g["grad/unused"] = np.asarray(no_grad)
        sd_params = dict(model.named_parameters())
        for n_ in ("input_proj.bias", "ground_decoder.decoder.bbox_embed.layers.2.weight", "temp_embed.layers.1.weight",
                   "ground_encoder.encoder.frame_cls.weight", "ground_encoder.encoder.video_cls.weight",
                   "ground_decoder.template_generator.anchor_proj.weight",
                   "ground_decoder.decoder.layers.0.ca_qpos_proj.bias",
                   "ground_decoder.temp_decoder.norm.weight",
                   "ground_encoder.encoder.spatial_layers.0.self_attn.in_proj_bias",
                   "vis_encoder.0.body.layer4.2.conv3.weight",
                   "vis_encoder.0.body.layer2.0.conv1.weight"):
            g[f"grad/full/{n_}"] = sub(sd_params[n_].grad)
    g["meta/state_dict_keys"] = np.asarray([k for k in model.state_dict().keys()
                                            if not k.startswith("text_encoder.")])
    g["meta/config"] = np.asarray([T, res, L], dtype=np.int64)
    return g


# ----------------------------------------------------------------------------
# model-level cases at every size the GPU tests use (round 4): the reference itself, fp32 and — for the cases with a
# backward — a second run in fp64, the "exact arithmetic" yardstick the calibrated gradient bound of
# tests/test_model_parity.py is built on.  SURVEY.md §8c: larger configs store outputs only; gradients are kept as a
# per-tensor sample (synth.sample_indices: <= 1024 elements) + the full-tensor norms.
# ----------------------------------------------------------------------------
def _run_reference_case(name: str, dtype: torch.dtype):
    T, res, L, pad, bwd = synth.MODEL_CASES[name]
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cfg, model, criterion, weight_dict, post, text = build_reference(L)
        from utils.misc import NestedTensor
        from utils.bounding_box import BoxList
        if dtype == torch.float64:
            model.double()
            (tm, tmem, tx), tcls = text
            model.text_encoder.text = ((tm, tmem.double(), tx), tcls.double())
            # PositionEmbeddingSine builds its table with dtype=torch.float32 whatever the default is
            # (vision_model/position_encoding.py:74-81): hand the SAME values on as fp64 so the fp64 Linear layers accept
            # them (a forward hook on the reference module; no reference code is changed)
            model.vis_encoder.register_forward_hook(lambda m, i, o: (o[0], o[1].double()))
        frames, mask, H, W = synth.synth_clip(T, res, pad)
        videos = NestedTensor(frames.to(dtype), mask, [T])
        act, boxes = synth.synth_targets(T)
        targets = [{"actioness": act, "boxs": BoxList(boxes.to(dtype), (W, H), mode="xyxy")}]
        stages = {}
        model.vis_encoder[0].body.layer4.register_forward_hook(lambda m, i, o: stages.__setitem__("layer4", o))
        model.ground_encoder.register_forward_hook(lambda m, i, o: stages.__setitem__("enc", o))
        model.ground_decoder.register_forward_hook(lambda m, i, o: stages.__setitem__("dec", o))
        with (torch.enable_grad() if bwd else torch.no_grad()):
            out = model(videos, ["synthetic query"])
        r = {"T": T, "H": H, "W": W, "L": L}
        r["out"] = {k: out[k].detach().clone() for k in ("pred_boxes", "pred_sted", "pred_actioness", "weights")}
        r["aux"] = [{k: a[k].detach().clone() for k in r["out"]} for a in out["aux_outputs"]]
        (hs, ref), (time_hs, weights) = stages["dec"]
        r["stage"] = {"layer4": stages["layer4"].detach(), "encoded_memory": stages["enc"]["encoded_memory"].detach(),
                      "frames_cls": stages["enc"]["frames_cls"].detach(), "hs": hs.detach(), "ref": ref.detach(),
                      "time_hs": time_hs.detach()}
        sizes = torch.tensor([[float(H), float(W)]], dtype=dtype).repeat(T, 1)
        pb, steds = post(out, sizes, [list(range(100, 100 + T))], [T])     # before the criterion edits pred_boxes in place
        r["post_boxes"], r["post_sted"] = pb.detach().clone(), steds
        if bwd:
            losses = criterion(out, targets, [T])
            assert set(losses.keys()) == set(weight_dict.keys())
            total = sum(losses[k] * weight_dict[k] for k in losses)
            total.backward()
            r["loss"] = {k: losses[k].item() for k in sorted(losses)}
            r["loss_total"] = total.item()
            r["grads"], r["unused"] = {}, []
            for n_, p in model.named_parameters():
                if n_.startswith("text_encoder."):
                    continue
                if p.grad is None:
                    if p.requires_grad:
                        r["unused"].append(n_)
                    continue
                r["grads"][n_] = p.grad.detach()
        return r
    finally:
        torch.set_default_dtype(prev)


def model_case_fixture(name: str):
    T, res, L, pad, bwd = synth.MODEL_CASES[name]
    r32 = _run_reference_case(name, torch.float32)
    g = {"meta/config": np.asarray([r32["T"], r32["H"], r32["W"], r32["L"]], dtype=np.int64),
         "meta/pad": np.asarray(pad or ""), "meta/backward": np.asarray(bool(bwd))}
    for k, v in r32["out"].items():
        g[f"out/{k}"] = v.numpy()
        for i, a in enumerate(r32["aux"]):
            g[f"out/aux{i}/{k}"] = a[k].numpy()
    g["post/boxes"] = r32["post_boxes"].numpy()
    g["post/sted"] = np.asarray(r32["post_sted"], dtype=np.int64)
    for k, v in r32["stage"].items():
        g[f"stage/{k}"] = sub(v, 1 << 13)
        g[f"stage/{k}/absmax"] = np.float32(v.abs().max().item())
    if not bwd:
        return g
    keys = sorted(r32["loss"])
    g["loss/keys"] = np.asarray(keys)
    g["loss/values"] = np.asarray([r32["loss"][k] for k in keys], dtype=np.float32)
    g["loss/total"] = np.float32(r32["loss_total"])
    g32 = r32["grads"]
    unused = r32["unused"]
    del r32
    import gc
    gc.collect()
    r64 = _run_reference_case(name, torch.float64)
    g64 = r64["grads"]
    assert list(g64) == list(g32) and r64["unused"] == unused
    g["loss/values64"] = np.asarray([r64["loss"][k] for k in keys], dtype=np.float64)
    g["loss/total64"] = np.float64(r64["loss_total"])
    for k, v in r64["out"].items():
        g[f"out64/{k}"] = v.numpy()
    g["post/sted64"] = np.asarray(r64["post_sted"], dtype=np.int64)
    names = list(g32)
    offs, s32, s64 = [0], [], []
    n32, n64, e32 = [], [], []
    for n_ in names:
        a, b = g32[n_].reshape(-1), g64[n_].reshape(-1)
        idx = torch.from_numpy(synth.sample_indices(n_, a.numel()))
        s32.append(a[idx].numpy().astype(np.float32))
        s64.append(b[idx].numpy().astype(np.float32))        # fp64 run, stored to fp32 precision (6e-8 relative)
        offs.append(offs[-1] + idx.numel())
        n32.append(a.double().norm().item())
        n64.append(b.norm().item())
        e32.append((a.double() - b).norm().item())
    g["grad/names"] = np.asarray(names)
    g["grad/unused"] = np.asarray(unused)
    g["grad/numel"] = np.asarray([g32[n_].numel() for n_ in names], dtype=np.int64)
    g["grad/offsets"] = np.asarray(offs, dtype=np.int64)
    g["grad/sample32"] = np.concatenate(s32)
    g["grad/sample64"] = np.concatenate(s64)
    g["grad/norm32"] = np.asarray(n32, dtype=np.float64)       # full-tensor L2 norms
    g["grad/norm64"] = np.asarray(n64, dtype=np.float64)
    g["grad/err32"] = np.asarray(e32, dtype=np.float64)        # || fp32 reference - fp64 reference ||, full tensor
    return g


def train_mode_fixture(name: str, trace_path: str):
    """model_<name>_train.npz: the TRAIN-mode step bench.py times, at its size, with given dropout masks.

    The reference draws its masks from torch's generator, which no other program can reproduce; the HIP path's masks are
    a pure function of (seed, device base + host offset + element index) (csrc/stcat_rng.h, host twin
    ops.dropout_keep_mask).  So the expected values come from the CPU ORACLE (oracle/stcat_oracle.py: pinned to the
    imported reference by the eval-mode fixtures of this directory, tests/test_oracle_golden.py) run in fp32 and fp64
    with the masks of a dropout stream recorded on the GPU box (tools/c3_train_trace.py -> `trace_path`: seed, base,
    (offset, decisions) per site).  The stream is stored in the fixture; the GPU test asserts that the live step draws
    exactly this stream before it compares anything.  Same arrays as model_case_fixture (minus the stage samples)."""
    import json
    from tests import test_model_parity as P
    T, res, L, pad, bwd = synth.MODEL_CASES[name]
    with open(trace_path) as f:
        tr = json.load(f)
    sites = [(int(o), int(n)) for o, n in tr["sites"]]
    seed, base = int(tr["seed"]), int(tr["base"])
    mk = lambda: P._HipMasks(sites, seed, base)   # noqa: E731
    frames, mask, H, W = synth.synth_clip(T, res, pad)
    out, boxes, sted, losses, g32 = P._run_oracle(T, res, L, True, torch.float32, pad, mk, trainable_only=True)
    g = {"meta/config": np.asarray([T, H, W, L], dtype=np.int64), "meta/pad": np.asarray(pad or ""),
         "meta/backward": np.asarray(True), "meta/source": np.asarray("oracle (train mode, masks of the recorded stream)"),
         "dropout/seed": np.asarray(seed, dtype=np.int64), "dropout/base": np.asarray(base, dtype=np.int64),
         "dropout/sites": np.asarray(sites, dtype=np.int64).reshape(-1, 2)}
    keys_o = ("pred_boxes", "pred_sted", "pred_actioness", "weights")
    for k in keys_o:
        g[f"out/{k}"] = out[k].detach().numpy()
        for i, a in enumerate(out["aux_outputs"]):
            g[f"out/aux{i}/{k}"] = a[k].detach().numpy()
    g["post/boxes"] = boxes.numpy()
    g["post/sted"] = np.asarray([sted], dtype=np.int64)
    keys = sorted(k for k in losses if k != "total")
    g["loss/keys"] = np.asarray(keys)
    g["loss/values"] = np.asarray([losses[k] for k in keys], dtype=np.float32)
    g["loss/total"] = np.float32(losses["total"])
    g32 = {k: v.detach().clone() for k, v in g32.items()}
    del out
    import gc
    gc.collect()
    out64, _, sted64, losses64, g64 = P._run_oracle(T, res, L, True, torch.float64, pad, mk, trainable_only=True)
    g["loss/values64"] = np.asarray([losses64[k] for k in keys], dtype=np.float64)
    g["loss/total64"] = np.float64(losses64["total"])
    for k in keys_o:
        g[f"out64/{k}"] = out64[k].detach().numpy()
    g["post/sted64"] = np.asarray([sted64], dtype=np.int64)
    names = [n_ for n_ in g32]
    assert set(g64) == set(g32)
    offs, s32, s64, n32, n64, e32 = [0], [], [], [], [], []
    for n_ in names:
        a, b = g32[n_].reshape(-1), g64[n_].reshape(-1)
        idx = torch.from_numpy(synth.sample_indices(synth.canonical_name(n_), a.numel()))
        s32.append(a[idx].numpy().astype(np.float32))
        s64.append(b[idx].numpy().astype(np.float32))
        offs.append(offs[-1] + idx.numel())
        n32.append(a.double().norm().item())
        n64.append(b.norm().item())
        e32.append((a.double() - b).norm().item())
    g["grad/names"] = np.asarray(names)
    g["grad/unused"] = np.asarray([], dtype=str)
    g["grad/numel"] = np.asarray([g32[n_].numel() for n_ in names], dtype=np.int64)
    g["grad/offsets"] = np.asarray(offs, dtype=np.int64)
    g["grad/sample32"] = np.concatenate(s32)
    g["grad/sample64"] = np.concatenate(s64)
    g["grad/norm32"] = np.asarray(n32, dtype=np.float64)
    g["grad/norm64"] = np.asarray(n64, dtype=np.float64)
    g["grad/err32"] = np.asarray(e32, dtype=np.float64)
    return g


def op_level_vectors():
    """Known-answer vectors for the small closed-form ops of the path."""
    install_stubs()
    sys.path.insert(0, REF)
    from models.net_utils import gen_sineembed_for_position, inverse_sigmoid
    from models.vision_model.position_encoding import PositionEmbeddingSine
    from models.grounding_model.position_encoding import SeqEmbeddingSine
    from models.grounding_model.attention import MultiheadAttention
    from models.post_processor import PostProcess
    from utils.misc import NestedTensor
    g = {}
    anchors = torch.from_numpy(synth.hash_uniform("op/anchors", 6 * 4).reshape(6, 1, 4)) * 0.5 + 0.5
    g["sine/anchors"] = anchors.numpy()
    g["sine/embed"] = gen_sineembed_for_position(anchors).numpy()
    x = torch.tensor([-0.5, 0.0, 1e-4, 1e-3, 0.25, 0.5, 0.999, 0.9995, 1.0, 1.5])
    g["invsig/x"] = x.numpy()
    g["invsig/y"] = inverse_sigmoid(x).numpy()
    # 2-D sine with a ragged pad mask
    m = torch.zeros(2, 5, 7, dtype=torch.bool)
    m[1, 3:, :] = True
    m[1, :, 5:] = True
    pe = PositionEmbeddingSine(128, normalize=True)
    g["pos2d/mask"] = m.numpy()
    g["pos2d/pos"] = pe(NestedTensor(torch.zeros(2, 1, 5, 7), m, [2])).numpy()
    g["seqsine/te"] = SeqEmbeddingSine(301, 256).te[:10].numpy()
    # DAB custom MHA: 1 query per batch row, k-dim 512, v-dim 256, partially masked keys
    torch.manual_seed(0)
    mha = MultiheadAttention(512, 8, dropout=0.0, vdim=256).eval()
    ow = torch.from_numpy(synth.synth_value("op/dab/out_proj.weight", (256, 256)).copy())
    ob = torch.from_numpy(synth.synth_value("op/dab/out_proj.bias", (256,)).copy())
    with torch.no_grad():
        mha.out_proj.weight.copy_(ow)
        mha.out_proj.bias.copy_(ob)
    q = torch.from_numpy(synth.hash_normal("op/dab/q", 3 * 512).reshape(1, 3, 512))
    k = torch.from_numpy(synth.hash_normal("op/dab/k", 11 * 3 * 512).reshape(11, 3, 512))
    v = torch.from_numpy(synth.hash_normal("op/dab/v", 11 * 3 * 256).reshape(11, 3, 256))
    kpm = torch.zeros(3, 11, dtype=torch.bool)
    kpm[1, 7:] = True
    kpm[2, 1:4] = True
    with torch.no_grad():
        o, _ = mha(q, k, v, key_padding_mask=kpm)
    g["dab/kpm"] = kpm.numpy()
    g["dab/out"] = o.numpy()
    # PostProcess temporal map: duration < T padding and an engineered near tie
    post = PostProcess()
    T = 12
    sted = torch.from_numpy(synth.hash_normal("op/post/sted", T * 2).reshape(1, T, 2)) * 2.0
    sted[0, 3, 0] = sted[0, :, 0].max() + 1.0
    sted[0, 7, 1] = sted[0, :, 1].max() + 1.0
    sted[0, 9, 1] = sted[0, 7, 1]  # exact tie between end=7 and end=9: first max wins
    boxes = torch.rand(T, 4, generator=torch.Generator().manual_seed(1)) * 0.5 + 0.25
    sizes = torch.tensor([[240.0, 320.0]]).repeat(T, 1)
    for dur in (12, 9, 6):
        pb, st = post({"pred_sted": sted, "pred_boxes": boxes}, sizes, [list(range(50, 50 + T))], [dur])
        g[f"post/dur{dur}/sted"] = np.asarray(st, dtype=np.int64)
    g["post/in_sted"] = sted.numpy()
    g["post/in_boxes"] = boxes.numpy()
    g["post/boxes"] = pb.numpy()
    return g


def eval_vectors():
    """Known answers of the evaluation tail (engine/evaluate.py:11-35 linear_interp, :111-119 span union)."""
    install_stubs()
    sys.path.insert(0, REF)
    from engine.evaluate import linear_interp
    g = {}
    # merged even/odd passes of a clip whose frame ids have gaps (sampled frames): ids and xyxy boxes
    ids = np.array([100, 101, 103, 104, 108, 109, 115], dtype=np.int64)
    boxes = (synth.hash_uniform("op/interp/boxes", len(ids) * 4).reshape(len(ids), 4) * 100 + 200).astype(np.float32)
    d = {int(f): [[float(v) for v in b]] for f, b in zip(ids, boxes)}
    out = linear_interp(dict(d))
    g["interp/ids"], g["interp/boxes"] = ids, boxes
    g["interp/out_ids"] = np.array(sorted(out), dtype=np.int64)
    g["interp/out_boxes"] = np.array([out[f][0] for f in sorted(out)], dtype=np.float64)
    one = linear_interp({7: [[1.0, 2.0, 3.0, 4.0]]})            # fewer than two frames: returned unchanged
    g["interp/single_ids"] = np.array(sorted(one), dtype=np.int64)
    return g


MAP2D_CFG = dict(MAX_MAP_SIZE=32, POOLING_COUNTS=[7, 4, 4], HIDDEN=64, HEADS=8, FFN_DIM=128, DROPOUT=0.1,
                 TEMP_PRED_LAYERS=2, TEMP_HEAD="conv", KERNAL_SIZE=3, CONV_LAYERS=2)


def map2d_vectors():
    """Known answers of models/map2d_head.py (Gen2DMap :9-62, TempPredictionHead 'conv' :65-127, TempConvInteraction
    :228-250) at a small configuration.  The reference hard-codes .to("cuda") (:33): patched to "cpu" for this run, as
    SURVEY.md §8c prescribes; the MODEL.TEMPFORMER config node does not exist in the reference's config and is supplied."""
    import importlib.util
    from types import SimpleNamespace as NS
    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        return orig_to(self, *tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a), **k)
    torch.Tensor.to = to_cpu
    try:
        spec = importlib.util.spec_from_file_location("ref_map2d_head", os.path.join(REF, "models", "map2d_head.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        cfg = NS(MODEL=NS(TEMPFORMER=NS(**MAP2D_CFG)))
        g = {}
        gen = m.Gen2DMap(cfg)
        g["map2d/mask"] = gen.mask2d.numpy()
        for tag, T in (("short", 20), ("long", 40)):     # T < N: adaptive max only; T > N: adaptive avg then max
            x = torch.from_numpy(synth.hash_normal(f"op/map2d/x{T}", 2 * T * 64).reshape(2, T, 64))
            g[f"map2d/{tag}/x"] = x.numpy()
            g[f"map2d/{tag}/map"] = gen(x).numpy()      # [b, d, N, N]
        head = m.TempPredictionHead(cfg).eval()
        with torch.no_grad():
            for k, v in head.state_dict().items():
                v.copy_(torch.from_numpy(synth.synth_value("map2d_head." + k, tuple(v.shape)).copy()))
            for i, w in enumerate(head.encoder.weights):
                g[f"map2d/weight{i}"] = w.numpy()
            xh = torch.from_numpy(synth.hash_normal("op/map2d/xh", 2 * 1 * 20 * 64).reshape(2, 1, 20, 64))
            g["map2d/head/x"] = xh.numpy()
            g["map2d/head/scores"] = head(xh).numpy()     # eval: sigmoid(scores) * mask  [layers, b, N, N]
        g["map2d/head/keys"] = np.array(list(head.state_dict().keys()))

        def fill(hd, prefix):
            with torch.no_grad():
                for k, v in hd.state_dict().items():
                    v.copy_(torch.from_numpy(synth.synth_value(prefix + k, tuple(v.shape)).copy()))

        def train_vectors(hd, x, tag, full_grads=True):
            """train mode (map2d_head.py:122-124: raw scores) + the gradients of sum(scores * G) — what a loss on the map
            would send back (round 3: the head's backward)"""
            hd.train()
            xr = x.clone().requires_grad_(True)
            sc = hd(xr)
            G = torch.from_numpy(synth.hash_normal(f"op/map2d/{tag}/G", sc.numel()).reshape(tuple(sc.shape)))
            g[f"map2d/{tag}/train_scores"] = sc.detach().numpy().copy()
            g[f"map2d/{tag}/G"] = G.numpy()
            (sc * G).sum().backward()
            g[f"map2d/{tag}/dx"] = xr.grad.numpy().copy()
            for k, prm in hd.named_parameters():
                if full_grads:
                    g[f"map2d/{tag}/grad/{k}"] = prm.grad.numpy().copy()
                else:
                    g[f"map2d/{tag}/gradnorm/{k}"] = np.array(float(prm.grad.double().norm()))

        # conv variant, train mode + backward (same weights / input as the eval vector above)
        train_vectors(head, xh, "head")
        # 'attn' variant (:130-205) at the model's width (HIDDEN 256, 8 heads: head dimension 32, LayerNorm(256)), small map,
        # dropout 0 so train mode is deterministic
        cfg_a = NS(MODEL=NS(TEMPFORMER=NS(**dict(MAP2D_CFG, TEMP_HEAD="attn", HIDDEN=256, HEADS=8, DROPOUT=0.0))))
        head_a = m.TempPredictionHead(cfg_a).eval()
        fill(head_a, "map2d_attn_head.")
        g["map2d/attn/keys"] = np.array(list(head_a.state_dict().keys()))
        xa = torch.from_numpy(synth.hash_normal("op/map2d/xa", 2 * 1 * 20 * 256).reshape(2, 1, 20, 256))
        g["map2d/attn/x"] = xa.numpy()
        with torch.no_grad():
            g["map2d/attn/scores"] = head_a(xa.clone()).numpy()
            head_a.train()     # raw scores (:122-124).  No gradient vector: the reference's `map2d[i] = encoder(map2d[i])`
            #                    (:113-115) overwrites a tensor its own backward needs — autograd refuses to differentiate it
            g["map2d/attn/train_scores"] = head_a(xa.clone()).numpy()
        # the reference-sized head: 128 x 128 map, 256 channels, four 9 x 9 convolutions
        cfg_f = NS(MODEL=NS(TEMPFORMER=NS(**dict(MAP2D_CFG, MAX_MAP_SIZE=128, POOLING_COUNTS=[15, 8, 8, 8], HIDDEN=256,
                                                  KERNAL_SIZE=9, CONV_LAYERS=4))))
        head_f = m.TempPredictionHead(cfg_f).eval()
        fill(head_f, "map2d_full_head.")
        xf = torch.from_numpy(synth.hash_normal("op/map2d/xf", 2 * 64 * 256).reshape(2, 1, 64, 256))  # (one map would be squeezed away, :119)
        g["map2d/full/x"] = xf.numpy()
        with torch.no_grad():
            g["map2d/full/scores"] = head_f(xf.clone()).numpy()
        train_vectors(head_f, xf, "full", full_grads=False)
    finally:
        torch.Tensor.to = orig_to
    return g


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    if sys.argv[1:] == ["map2d"]:
        np.savez_compressed(os.path.join(out_dir, "map2d.npz"), **map2d_vectors())
        print("map2d.npz written")
        return
    if sys.argv[1:] == ["eval"]:
        np.savez_compressed(os.path.join(out_dir, "eval.npz"), **eval_vectors())
        print("eval.npz written")
        return
    torch.set_num_threads(8)
    if sys.argv[1:2] == ["train"]:
        # python tests/golden/make_golden.py train C3 gpurun_out/c3_train_trace.json -> tests/golden/model_C3_train.npz
        import shutil
        import time
        name, trace_path = sys.argv[2], sys.argv[3]
        t0 = time.time()
        g = train_mode_fixture(name + "_train", trace_path)
        path = os.path.join(out_dir, f"model_{name}_train.npz")
        np.savez_compressed(path, **g)
        shutil.copyfile(trace_path, os.path.join(out_dir, f"model_{name}_train_trace.json"))
        print(f"model_{name}_train.npz written: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s",
              flush=True)
        return
    if sys.argv[1:2] == ["model"]:
        # python tests/golden/make_golden.py model [CASE ...]  -> tests/golden/model_<CASE>.npz (C3: ~10 minutes, 40 GB)
        import time
        for name in (sys.argv[2:] or list(synth.MODEL_CASES)):
            t0 = time.time()
            g = model_case_fixture(name)
            path = os.path.join(out_dir, f"model_{name}.npz")
            np.savez_compressed(path, **g)
            print(f"model_{name}.npz written: {len(g)} arrays, {os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s",
                  flush=True)
        return
    np.savez_compressed(os.path.join(out_dir, "ops.npz"), **op_level_vectors())
    print("ops.npz written")
    which = sys.argv[1:] or ["C1"]
    for name in which:
        g = run_config(name, with_backward=(name == "C1"))
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **g)
        print(name, "written:", len(g), "arrays")


if __name__ == "__main__":
    main()
