"""Deterministic, version-independent synthetic weights and inputs.

The reference checkpoints (3.1 GB, README.md:129-135) and the datasets cannot
travel, so parity is established on synthetic weights that are a pure function
of the *parameter name* and the flat element index (SURVEY.md §8c "Weights for
parity").  Nothing here depends on ``torch.manual_seed`` streams, the torch
version or the device: values come from a 64-bit counter hash evaluated with
numpy, so the golden-fixture script (which fills the imported reference
modules), the CPU oracle and the GPU box all regenerate identical tensors from
names alone.

State-dict naming follows the reference module tree (SURVEY.md §8b
"Parameter / checkpoint compatibility").
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

_MASK64 = (1 << 64) - 1
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & _MASK64
    return h


def _splitmix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def hash_uniform(name: str, n: int, salt: int = 0) -> np.ndarray:
    """n float32 values in [-1, 1), a pure function of (name, salt, index)."""
    seed = np.uint64((fnv1a64(name) + 0x632BE59BD9B4E019 * (salt + 1)) & _MASK64)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = _splitmix(idx * _GOLDEN + seed)
    top = (x >> np.uint64(40)).astype(np.float64)  # 24 random bits
    return (top / float(1 << 23) - 1.0).astype(np.float32)


def hash_normal(name: str, n: int, salt: int = 0) -> np.ndarray:
    """Approximately N(0,1) float32 (sum of 4 uniforms, variance-normalised)."""
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        acc += hash_uniform(name, n, salt=salt * 4 + k + 101).astype(np.float64)
    return (acc * math.sqrt(3.0 / 4.0)).astype(np.float32)


# --------------------------------------------------------------------------
# parameter shapes of the hot path (reference module tree)
# --------------------------------------------------------------------------
RESNET101_BLOCKS = (3, 4, 23, 3)
RESNET_PLANES = (64, 128, 256, 512)


def backbone_entries(prefix: str = "vis_encoder.0.body.") -> List[Tuple[str, Tuple[int, ...]]]:
    """torchvision ResNet-101 v1.5 keys as wrapped by BackboneBase
    (models/vision_model/backbone.py:69-121); BN entries are buffers."""
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def bn(name: str, c: int):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            out.append((f"{prefix}{name}.{leaf}", (c,)))

    out.append((prefix + "conv1.weight", (64, 3, 7, 7)))
    bn("bn1", 64)
    inplanes = 64
    for li, (nblk, planes) in enumerate(zip(RESNET101_BLOCKS, RESNET_PLANES), start=1):
        for bi in range(nblk):
            base = f"layer{li}.{bi}."
            out.append((prefix + base + "conv1.weight", (planes, inplanes, 1, 1)))
            bn(base + "bn1", planes)
            out.append((prefix + base + "conv2.weight", (planes, planes, 3, 3)))
            bn(base + "bn2", planes)
            out.append((prefix + base + "conv3.weight", (planes * 4, planes, 1, 1)))
            bn(base + "bn3", planes * 4)
            if bi == 0:
                out.append((prefix + base + "downsample.0.weight", (planes * 4, inplanes, 1, 1)))
                bn(base + "downsample.1", planes * 4)
            inplanes = planes * 4
    return out


def _mha_entries(p: str, d: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [
        (p + "in_proj_weight", (3 * d, d)),
        (p + "in_proj_bias", (3 * d,)),
        (p + "out_proj.weight", (d, d)),
        (p + "out_proj.bias", (d,)),
    ]


def _lin(p: str, out_f: int, in_f: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [(p + "weight", (out_f, in_f)), (p + "bias", (out_f,))]


def _ln(p: str, d: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [(p + "weight", (d,)), (p + "bias", (d,))]


def grounding_entries(d: int = 256, ffn: int = 2048, n_enc: int = 6, n_dec: int = 6,
                      max_len: int = 300, query_dim: int = 4) -> List[Tuple[str, Tuple[int, ...]]]:
    """input_proj + ground_encoder + ground_decoder + heads
    (models/pipeline.py:33-50, modal_encoder.py:11-128, query_decoder.py:13-81)."""
    e: List[Tuple[str, Tuple[int, ...]]] = []
    e += [("input_proj.weight", (d, 2048, 1, 1)), ("input_proj.bias", (d,))]
    enc = "ground_encoder.encoder."
    for kind in ("spatial_layers", "temporal_layers"):
        for i in range(n_enc):
            p = f"{enc}{kind}.{i}."
            e += _mha_entries(p + "self_attn.", d)
            e += _lin(p + "linear1.", ffn, d) + _lin(p + "linear2.", d, ffn)
            e += _ln(p + "norm1.", d) + _ln(p + "norm2.", d)
    e += [(enc + "time_embed.te", (max_len + 1, 1, d))]
    e += [(enc + "local_pos_embed.weight", (1, d)), (enc + "frame_cls.weight", (1, d)),
          (enc + "video_cls.weight", (1, d))]
    e += _lin("ground_encoder.fusion.", d, d)
    dec = "ground_decoder."
    for nm in ("content_proj", "gamma_proj", "beta_proj"):
        e += _lin(f"{dec}template_generator.{nm}.", d, d)
    e += _lin(f"{dec}template_generator.anchor_proj.", query_dim, d)
    for i in range(n_dec):
        p = f"{dec}decoder.layers.{i}."
        for nm in ("sa_qcontent_proj", "sa_qpos_proj", "sa_qtime_proj", "sa_kcontent_proj",
                   "sa_kpos_proj", "sa_ktime_proj", "sa_v_proj"):
            e += _lin(p + nm + ".", d, d)
        e += _mha_entries(p + "self_attn.", d)
        e += _lin(p + "ca_qcontent_proj.", d, d)
        if i == 0:  # query_decoder.py:166-167
            e += _lin(p + "ca_qpos_proj.", d, d)
        for nm in ("ca_kcontent_proj", "ca_kpos_proj", "ca_qtime_proj", "ca_v_proj", "ca_qpos_sine_proj"):
            e += _lin(p + nm + ".", d, d)
        e += _lin(p + "cross_attn.out_proj.", d, d)
        e += _lin(p + "linear1.", ffn, d) + _lin(p + "linear2.", d, ffn)
        e += _ln(p + "norm1.", d) + _ln(p + "norm3.", d) + _ln(p + "norm4.", d)
    e += _ln(dec + "decoder.norm.", d)
    e += _lin(dec + "decoder.query_scale.layers.0.", d, d) + _lin(dec + "decoder.query_scale.layers.1.", d, d)
    e += _lin(dec + "decoder.ref_point_head.layers.0.", d, query_dim // 2 * d)
    e += _lin(dec + "decoder.ref_point_head.layers.1.", d, d)
    # decoder.bbox_embed aliases the top-level bbox_embed (pipeline.py:50)
    for j, (o, i_) in enumerate(((d, d), (d, d), (4, d))):
        e += _lin(f"{dec}decoder.bbox_embed.layers.{j}.", o, i_)
    for i in range(n_dec):
        p = f"{dec}temp_decoder.layers.{i}."
        e += _mha_entries(p + "self_attn.", d) + _mha_entries(p + "cross_attn_image.", d)
        e += _lin(p + "linear1.", ffn, d) + _lin(p + "linear2.", d, ffn)
        e += _ln(p + "norm1.", d) + _ln(p + "norm3.", d) + _ln(p + "norm4.", d)
    e += _ln(dec + "temp_decoder.norm.", d)
    e += [(dec + "time_embed.te", (max_len + 1, 1, d))]
    e += _lin("temp_embed.layers.0.", d, d) + _lin("temp_embed.layers.1.", 2, d)
    for j, (o, i_) in enumerate(((d, d), (d, d), (4, d))):
        e += _lin(f"bbox_embed.layers.{j}.", o, i_)
    e += _lin("action_embed.layers.0.", d, d) + _lin("action_embed.layers.1.", 1, d)
    return e


def hot_path_entries(**kw) -> List[Tuple[str, Tuple[int, ...]]]:
    return backbone_entries() + grounding_entries(**kw)


_ALIAS = re.compile(r"^ground_decoder\.decoder\.bbox_embed\.")


def canonical_name(name: str) -> str:
    """The decoder's bbox_embed is the same module object as the top-level one
    (models/pipeline.py:50); both keys must map to the same values."""
    return _ALIAS.sub("bbox_embed.", name)


def time_sine_table(rows: int, d: int = 256) -> np.ndarray:
    """SeqEmbeddingSine buffer (models/grounding_model/position_encoding.py:23-33)."""
    pos = np.arange(rows, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-math.log(10000.0) / d)).astype(np.float32)
    te = np.zeros((rows, 1, d), dtype=np.float32)
    ang = (pos * div[None, :]).astype(np.float32)
    te[:, 0, 0::2] = np.sin(ang)
    te[:, 0, 1::2] = np.cos(ang)
    return te


_VALUE_CACHE: Dict[Tuple[str, Tuple[int, ...]], np.ndarray] = {}
_VALUE_CACHE_LIMIT = 1 << 28      # elements (1 GB of fp32): the 82 M-parameter model three times over


def synth_value(name: str, shape: Tuple[int, ...]) -> np.ndarray:
    """Value of one state-dict entry, from its name alone.  Memoised (read-only arrays): a test process builds the
    82 M-parameter model dozens of times and the counter hash costs ~3 s per model."""
    key = (canonical_name(name), tuple(int(d) for d in shape))
    v = _VALUE_CACHE.get(key)
    if v is None:
        v = _synth_value(*key)
        v.setflags(write=False)
        if sum(a.size for a in _VALUE_CACHE.values()) + v.size <= _VALUE_CACHE_LIMIT:
            _VALUE_CACHE[key] = v
    return v


def _synth_value(name: str, shape: Tuple[int, ...]) -> np.ndarray:
    name = canonical_name(name)
    n = int(np.prod(shape))
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "te":
        return time_sine_table(shape[0], shape[2])
    u = hash_uniform(name, n).reshape(shape)
    parent = name.rsplit(".", 1)[0]
    is_bn = bool(re.search(r"(\.bn\d|downsample\.1|^vis_encoder\.0\.body\.bn1)$", parent))
    if is_bn:
        if leaf == "weight":
            w = 1.0 + 0.5 * u
            if re.search(r"\.bn3$", parent):  # last BN of a bottleneck: keep 33 residual adds O(1)
                w = w * 0.2
            return w.astype(np.float32)
        if leaf == "bias":
            return (0.1 * u).astype(np.float32)
        if leaf == "running_mean":
            return (0.1 * u).astype(np.float32)
        if leaf == "running_var":
            return (1.0 + 0.5 * u).astype(np.float32)
    if re.search(r"\.norm\d?$", parent) or parent.endswith(".norm"):
        return (1.0 + 0.1 * u if leaf == "weight" else 0.05 * u).astype(np.float32)
    if len(shape) == 4:  # conv weights: kaiming-uniform, relu gain
        fan_in = shape[1] * shape[2] * shape[3]
        return (u * math.sqrt(6.0 / fan_in)).astype(np.float32)
    if len(shape) == 2 and leaf in ("weight", "in_proj_weight"):
        if shape[0] == 1:  # nn.Embedding(1, d) tokens
            return u.astype(np.float32)
        fan_out, fan_in = shape
        if leaf == "in_proj_weight":
            fan_out = fan_out // 3
        return (u * math.sqrt(6.0 / (fan_in + fan_out))).astype(np.float32)
    # biases
    return (0.05 * u).astype(np.float32)


def synth_state_dict(entries: Iterable[Tuple[str, Tuple[int, ...]]] | None = None,
                     device: str | torch.device = "cpu", **kw) -> Dict[str, torch.Tensor]:
    if entries is None:
        entries = hot_path_entries(**kw)
    return {k: torch.from_numpy(synth_value(k, s).copy()).to(device) for k, s in entries}


def fill_module_(module: torch.nn.Module, skip_prefixes: Tuple[str, ...] = ("text_encoder.",)) -> List[str]:
    """Overwrite every parameter/buffer of ``module`` in place with its synthetic value."""
    filled = []
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if k.startswith(skip_prefixes) or not v.dtype.is_floating_point:
                continue
            v.copy_(torch.from_numpy(synth_value(k, tuple(v.shape)).copy()).to(v.device))
            filled.append(k)
    return filled


# --------------------------------------------------------------------------
# synthetic inputs (SURVEY.md §8d "Synthetic inputs")
# --------------------------------------------------------------------------
CONFIGS = {
    # name: (T, resolution, text tokens)
    "C1": (8, 224, 10),
    "C2": (32, 416, 10),
    "C3": (64, 448, 10),
    "C5": (128, 448, 40),
}


def synth_frames(T: int, res: int, seed: int = 0) -> torch.Tensor:
    x = hash_normal(f"frames/{T}/{res}", T * 3 * res * res, salt=seed)
    return torch.from_numpy(x.reshape(T, 3, res, res))


def synth_text(L: int, d: int = 256, seed: int = 0):
    """Boundary tensors the (out-of-scope) text encoder hands to the hot path:
    (mask[1,L] bool, memory[L,1,d], None), cls[1,d]  (language_model/bert.py:59-74)."""
    mem = hash_normal(f"text/mem/{L}", L * d, salt=seed).reshape(L, 1, d)
    mem = (mem - mem.mean(-1, keepdims=True)) / np.sqrt(mem.var(-1, keepdims=True) + 1e-12)
    cls = hash_normal(f"text/cls/{L}", d, salt=seed).reshape(1, d)
    mask = torch.zeros(1, L, dtype=torch.bool)
    return (mask, torch.from_numpy(mem.astype(np.float32)), None), torch.from_numpy(cls)


def synth_targets(T: int, seed: int = 0):
    """actioness = 1 on [T/4, 3T/4); boxes cxcywh (0.5,0.5,0.2,0.3) jittered +-0.05."""
    act = torch.zeros(T, dtype=torch.long)
    s, e = T // 4, (3 * T) // 4
    act[s:e] = 1
    jit = hash_uniform(f"targets/{T}", (e - s) * 4, salt=seed).reshape(e - s, 4) * 0.05
    boxes = torch.tensor([0.5, 0.5, 0.2, 0.3]) + torch.from_numpy(jit)
    return act, boxes.float()


def synth_clip(T: int, res, pad=None):
    """frames [T,3,H,W] + padding mask [T,H,W]; res = side of a square clip or (H, W).  pad="ragged": frames of
    different extents inside one padded tensor, as NestedTensor.from_tensor_list builds them (utils/misc.py:67-94: zeros
    + mask = True): the last frame loses its right quarter, frame 1 its bottom eighth, frame 2 both."""
    H, W = (res, res) if isinstance(res, int) else res
    frames = synth_frames(T, max(H, W))[:, :, :H, :W].contiguous()
    mask = torch.zeros(T, H, W, dtype=torch.bool)
    if pad == "ragged":
        mask[T - 1, :, W - W // 4:] = True
        mask[1 % T, H - H // 8:, :] = True
        mask[2 % T, H - H // 8:, :] = True
        mask[2 % T, :, W - W // 4:] = True
        frames = frames.masked_fill(mask[:, None], 0.0)
    return frames, mask, H, W


def sample_indices(name: str, numel: int, k: int = 1024) -> np.ndarray:
    """Up to k distinct flat indices into a tensor of `numel` elements, a pure function of (name, numel, k): the
    gradient fixtures keep these elements of every parameter gradient (a flat stride would always hit the same filter
    tap / input channel of a conv weight).  Sorted, int64."""
    if numel <= k:
        return np.arange(numel, dtype=np.int64)
    seed = np.uint64(fnv1a64("sample/" + canonical_name(name)))
    with np.errstate(over="ignore"):
        x = _splitmix(np.arange(2 * k, dtype=np.uint64) * _GOLDEN + seed)
    idx = np.unique((x % np.uint64(numel)).astype(np.int64))
    if idx.size > k:
        idx = idx[np.linspace(0, idx.size - 1, k).astype(np.int64)]
    return idx


# model-level parity cases: name -> (T, resolution or (H, W), text tokens, padding, with backward).  The fixtures
# tests/golden/model_<name>.npz hold what the imported reference computes for them (tests/golden/make_golden.py).
MODEL_CASES = {
    "C1": (8, 224, 10, None, True),
    "C2": (32, 416, 10, None, False),
    "C3": (64, 448, 10, None, True),
    "C5": (128, 448, 40, None, False),
    "SQ8_ragged": (8, 224, 10, "ragged", True),
    "NS8": (8, (405, 720), 10, None, True),
    "NS8_ragged": (8, (405, 720), 10, "ragged", True),
    # train-mode fixtures (round 6): the same clips, expected values from the oracle fed with the recorded dropout stream
    # of the benchmark's own step (tests/golden/make_golden.py train ...; files model_<case>.npz)
    "C1_train": (8, 224, 10, None, True),
    "C3_train": (64, 448, 10, None, True),
}
