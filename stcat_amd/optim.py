"""Optimizer tail of the training step as two multi-tensor HIP launches.

Mirrors the reference's interface for this part of the loop (scripts/train_net.py:134-143):

    optimizer = make_optimizer(cfg, model)                      # engine/optimizer.py:25-55 (AdamW, 4 groups)
    ...
    optimizer.zero_grad(); losses.backward()
    optimizer.step(max_grad_norm=cfg.SOLVER.MAX_GRAD_NORM,      # clip_grad_norm_ + AdamW.step + update_ema
                   model_ema=model_ema, ema_decay=cfg.MODEL.EMA_DECAY)
    adjust_learning_rate(cfg, optimizer, iteration, max_iter)   # engine/lr_scheduler.py:212-252

`clip_grad_norm_` and `update_ema` are also available on their own with the reference's signatures.  There is no
PyTorch fallback: every update runs in csrc/optim.h through the C ABI.
"""
from bisect import bisect_right
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from . import _lib as L

CHUNK = 1 << 16  # elements per workgroup


class _TensorTable:
    """Device table of (p, g, m, v, ema, n, group) entries + the chunk lists of csrc/optim.h.  Rebuilt only when a
    pointer changes (gradient tensors are re-allocated by autograd unless they live in the reducer's flat buckets
    or the zero arena, where they are stable from step to step)."""

    def __init__(self, device):
        self.device = device
        self.key = None
        self.table = self.chunk_tensor = self.chunk_off = None
        self.n_chunks = 0
        self.entry = L.load().stcat_optim_table_entry_bytes()
        assert self.entry == 56, self.entry

    def update(self, rows):
        """rows: list of (p_ptr, g_ptr, m_ptr, v_ptr, ema_ptr, numel, group)"""
        key = tuple(rows)
        if key == self.key:
            return
        self.key = key
        dt = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("ema", "<u8"), ("n", "<i8"),
                       ("group", "<i4"), ("pad", "<i4")])
        assert dt.itemsize == self.entry
        tab = np.zeros(len(rows), dtype=dt)
        ct, co = [], []
        for i, (p, g, m, v, e, n, grp) in enumerate(rows):
            tab[i] = (p, g, m, v, e, n, grp, 0)
            for off in range(0, n, CHUNK):
                ct.append(i)
                co.append(off)
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32).to(self.device)
        self.chunk_off = torch.tensor(co, dtype=torch.int64).to(self.device)
        self.n_chunks = len(ct)


def _dense(t: torch.Tensor) -> bool:
    """element-wise kernels only need a gap-free buffer: row-major, or channels_last (the conv weights' OHWI)"""
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """same element order in memory (strides of size-1 dims are arbitrary and ignored)"""
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


class AdamW:
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt) with the
    gradient-norm clip and the EMA model update folded into the same pass.  `param_groups` is a list of dicts with
    "params", "lr", "weight_decay" (what adjust_learning_rate and checkpoint code touch)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{"params": groups}]
        self.param_groups: List[Dict] = []
        for g in groups:
            g = dict(g)
            g["params"] = [p for p in g["params"]]
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            self.param_groups.append(g)
        if len(self.param_groups) > 8:
            raise ValueError("at most 8 parameter groups")
        self.betas, self.eps = betas, eps
        self.state: Dict[torch.Tensor, Dict[str, torch.Tensor]] = {}
        self.step_count = 0
        self._table = None
        self._sq = None
        self._plist = [p for g in self.param_groups for p in g["params"]]
        self._gkey = None
        self._any = None

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def _rows(self, ema_of):
        rows = []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not _dense(p) or not _same_layout(p.grad, p):
                    raise L.StcatHipError("optimizer tensors must be dense and share one memory layout")
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = {"exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                e = ema_of.get(p) if ema_of else None
                if e is not None and not _same_layout(e, p):
                    raise L.StcatHipError("EMA tensor layout differs from its parameter")
                rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             _ptr(e), p.numel(), gi))
        return rows

    @torch.no_grad()
    def step(self, max_grad_norm: float = 0.0, model_ema=None, ema_decay: float = 0.0, model=None):
        """One optimizer step.  max_grad_norm > 0 clips by the global norm first (train_net.py:136-137); with
        `model_ema` (and the `model` it shadows) the EMA weights are updated in the same launch
        (engine/optimizer.py:5-22).  Returns the squared gradient norm as a device scalar (no sync)."""
        plist = self._plist
        # cheap staleness key: the gradient pointers (everything else in the table is stable between steps)
        gkey = (id(model_ema),) + tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in plist)
        if gkey != self._gkey:
            ema_of = None
            if model_ema is not None:
                if model is None:
                    raise ValueError("model_ema needs the model it tracks")
                ema_sd = dict(model_ema.named_parameters())
                ema_of = {p: ema_sd[n] for n, p in model.named_parameters() if n in ema_sd}
            rows = self._rows(ema_of)
            if not rows:
                return None
            self._any = next(p for p in plist if p.grad is not None)
            if self._table is None:
                self._table = _TensorTable(self._any.device)
                self._sq = torch.zeros(1, device=self._any.device)
            self._table.update(rows)
            self._gkey = gkey
        any_p = self._any
        t = self._table
        stream = L.stream_of(any_p)
        self.step_count += 1
        from . import ops
        ops.WEIGHT_EPOCH += 1  # parameters change below through raw pointers: derived copies (weight planes) are stale
        sq = None
        if max_grad_norm and max_grad_norm > 0:
            L.call("stcat_grad_sqnorm", t.table.data_ptr(), t.chunk_tensor.data_ptr(), t.chunk_off.data_ptr(),
                   t.n_chunks, CHUNK, self._sq.data_ptr(), stream)
            sq = self._sq
        lr = np.array([g["lr"] for g in self.param_groups], dtype=np.float32)
        wd = np.array([g["weight_decay"] for g in self.param_groups], dtype=np.float32)
        L.call("stcat_adamw_ema_step", t.table.data_ptr(), t.chunk_tensor.data_ptr(), t.chunk_off.data_ptr(),
               t.n_chunks, CHUNK, _ptr(sq), lr.ctypes.data, wd.ctypes.data, len(self.param_groups),
               float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_count,
               float(max_grad_norm or 0.0), float(ema_decay), stream)
        return sq

    def state_dict(self):
        """torch.optim.AdamW's layout (what the reference's Checkpointer round-trips, utils/checkpoint.py):
        state[idx] = {step, exp_avg, exp_avg_sq}, param_groups[i]["params"] = [idx, ...]."""
        idx, groups = {}, []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                idx[p] = len(idx)
                ids.append(idx[p])
            d = {k: v for k, v in g.items() if k != "params"}
            d.setdefault("amsgrad", False)
            d["params"] = ids
            groups.append(d)
        state = {}
        for p, st in self.state.items():
            state[idx[p]] = {"step": torch.tensor(float(st.get("step", self.step_count))),
                             "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """accepts a torch.optim.AdamW checkpoint (per-parameter `step`) as well as its own"""
        params = [p for g in self.param_groups for p in g["params"]]
        saved_ids = [i for g in sd["param_groups"] for i in g["params"]] if sd["param_groups"] and \
            "params" in sd["param_groups"][0] else list(range(len(params)))
        if len(saved_ids) != len(params):
            raise ValueError("loaded state dict has a different number of parameters")
        where = {sid: p for sid, p in zip(saved_ids, params)}
        steps = []
        self.state = {}
        for i, st in sd["state"].items():
            p = where[int(i)]
            # moments take the parameter's own memory layout (channels_last conv weights); copy_ maps by index
            self.state[p] = {k: torch.zeros_like(p).copy_(st[k].to(p.device, torch.float32))
                             for k in ("exp_avg", "exp_avg_sq")}
            if "step" in st:
                steps.append(int(float(st["step"])))
        # one bias-correction counter for the fused launch: every parameter of the hot path receives a gradient at
        # every step (statically dead ones never enter `state`), so the per-parameter counters agree
        self.step_count = max(steps) if steps else int(sd.get("step", 0))
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})
        self._gkey = None


def make_optimizer(cfg, model, logger=None) -> AdamW:
    """engine/optimizer.py:25-55: four groups (rest / vis_encoder / text_encoder / ground_decoder.temp_decoder) with
    their own learning rates; only SOLVER.OPTIMIZER == 'adamw' (both experiment files) is implemented."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    vis = [p for n, p in named if "vis_encoder" in n]
    txt = [p for n, p in named if "text_encoder" in n]
    tmp = [p for n, p in named if "ground_decoder.temp_decoder" in n]
    rest = [p for n, p in named if "vis_encoder" not in n and "text_encoder" not in n
            and "ground_decoder.temp_decoder" not in n]
    if cfg.SOLVER.OPTIMIZER != "adamw":
        raise ValueError("stcat_amd.optim implements SOLVER.OPTIMIZER='adamw' only")
    groups = [{"params": rest}, {"params": vis, "lr": cfg.SOLVER.VIS_BACKBONE_LR},
              {"params": txt, "lr": cfg.SOLVER.TEXT_LR}, {"params": tmp, "lr": cfg.SOLVER.TEMP_LR}]
    return AdamW(groups, lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY)


def adjust_learning_rate(cfg, optimizer, curr_step: int, num_training_steps: int) -> None:
    """engine/lr_scheduler.py:212-252 (host-side scalars: the fused step reads the group learning rates by value)."""
    warm = round(cfg.SOLVER.WARMUP_PROP * num_training_steps)
    per_epoch = round(num_training_steps / cfg.SOLVER.MAX_EPOCH)
    epoch = curr_step // per_epoch
    kind = cfg.SOLVER.SCHEDULE.TYPE
    stepped = 0.1 ** bisect_right(cfg.SOLVER.SCHEDULE.DROP_STEP, epoch)
    if kind == "multistep_with_warmup":
        gamma = stepped
        if curr_step < warm:
            side = float(curr_step) / float(max(1, warm))
        else:
            side = max(0.0, float(num_training_steps - curr_step) / float(max(1, num_training_steps - warm)))
    elif kind == "multistep_with_warmup_all":
        gamma = float(curr_step) / float(max(1, warm)) if curr_step < warm else stepped
        side = gamma
    else:
        raise ValueError(f"Unsupported Schedule Type : {kind}")
    base = [cfg.SOLVER.BASE_LR, cfg.SOLVER.VIS_BACKBONE_LR, cfg.SOLVER.TEXT_LR, cfg.SOLVER.TEMP_LR]
    assert len(optimizer.param_groups) == len(base)
    for group, lr, gm in zip(optimizer.param_groups, base, [gamma, gamma, side, side]):
        group["lr"] = lr * gm


@torch.no_grad()
def clip_grad_norm_(parameters: Iterable[torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (L2): returns the total norm (device scalar) and scales the gradients in
    place when it exceeds max_norm.  The fused AdamW.step(max_grad_norm=...) does not need this call.
    Both launches walk every gradient buffer in flat memory order (the ops are element-wise), so channels_last
    conv gradients — the layout of every 3x3 weight gradient of the backbone — are handled without a layout copy."""
    ps = [p for p in parameters if p.grad is not None]
    if not ps:
        return torch.zeros(())
    for p in ps:
        L.check_tensor(p.grad, "gradient")
        if not _dense(p.grad):
            raise L.StcatHipError("clip_grad_norm_: gradients must be dense (row-major or channels_last)")
    dev = ps[0].device
    tab = _TensorTable(dev)
    tab.update([(0, p.grad.data_ptr(), 0, 0, 0, p.grad.numel(), 0) for p in ps])
    sq = torch.zeros(1, device=dev)
    st = L.stream_of(ps[0])
    L.call("stcat_grad_sqnorm", tab.table.data_ptr(), tab.chunk_tensor.data_ptr(), tab.chunk_off.data_ptr(),
           tab.n_chunks, CHUNK, sq.data_ptr(), st)
    # g *= min(1, max_norm / (norm + 1e-6)) with the coefficient computed on the device: no host sync
    L.call("stcat_grad_clip_scale", tab.table.data_ptr(), tab.chunk_tensor.data_ptr(), tab.chunk_off.data_ptr(),
           tab.n_chunks, CHUNK, sq.data_ptr(), float(max_norm), st)
    return sq.sqrt()[0]


@torch.no_grad()
def update_ema(model, model_ema, decay: float) -> None:
    """engine/optimizer.py:5-22 on its own: every floating-point state_dict entry of model_ema moves toward the
    model's (integer buffers are copied)."""
    if hasattr(model, "module"):
        model = model.module
    msd = model.state_dict()
    rows, dev, any_t = [], None, None
    for k, ema_v in model_ema.state_dict().items():
        src = msd[k].detach()
        if ema_v.dtype != torch.float32:
            ema_v.copy_(src)
            continue
        if not _dense(ema_v) or not _same_layout(src, ema_v):
            raise L.StcatHipError("update_ema: tensors must be dense and share one memory layout")
        rows.append((src.data_ptr(), 0, 0, 0, ema_v.data_ptr(), ema_v.numel(), 0))
        dev, any_t = ema_v.device, ema_v
    if not rows:
        return
    tab = _TensorTable(dev)
    tab.update(rows)
    L.call("stcat_ema_update", tab.table.data_ptr(), tab.chunk_tensor.data_ptr(), tab.chunk_off.data_ptr(),
           tab.n_chunks, CHUNK, float(decay), L.stream_of(any_t))



def set_f16_scales(weight_log2: int, grad_log2: int) -> None:
    """the only supported way to change mode f16x3p's operand scales: a new WEIGHT scale makes every cached weight plane
    stale (they hold w * 2^old), so the weight-plane caches are forced to refresh (ops.WEIGHT_EPOCH) and every launch plan
    is dropped (the caches are keyed on pointers + mode, not on the scales)"""
    from . import ops, plans
    old_w = L.load().stcat_get_f16_scale(0)
    L.call("stcat_set_f16_scales", int(weight_log2), int(grad_log2))
    if int(weight_log2) != int(old_w):
        ops.WEIGHT_EPOCH += 1
    plans.invalidate()


class PlaneLossScale:
    """Dynamic gradient (loss) scale of the experimental mma mode `f16x3p` — torch.cuda.amp.GradScaler's policy on the one
    number that mode needs: every GRADIENT plane of the backbone holds dy * 2^log2 (fp16's range; csrc/igemm_pl.h), the
    weight-gradient epilogue divides it out again, so the scale never reaches the optimizer.  An overflow shows up as a
    non-finite global gradient norm: `AdamW.step(max_grad_norm=...)` then skips the update on the device, and `update()`
    halves the scale; after `growth_interval` finite steps in a row it doubles (up to `max_log2`).  `update()` reads one
    device scalar (a sync) every `check_every` steps.  Changing the scale invalidates the launch plans (they bake the
    epilogue factor into their recorded arguments)."""

    def __init__(self, init_log2: int = 16, growth_interval: int = 2000, check_every: int = 1, max_log2: int = 24):
        self.log2 = int(init_log2)
        self.growth_interval = int(growth_interval)
        self.check_every = max(1, int(check_every))
        self.max_log2 = int(max_log2)
        self.good = 0
        self.calls = 0
        self.skipped = 0
        self._apply()

    def _apply(self):
        from . import plans
        wlog = L.load().stcat_get_f16_scale(0)
        set_f16_scales(int(wlog), int(self.log2))

    def update(self, sqnorm) -> bool:
        """sqnorm: what AdamW.step() returned.  -> True when the step was applied (finite norm)"""
        self.calls += 1
        if sqnorm is None or self.calls % self.check_every:
            return True
        ok = bool(torch.isfinite(sqnorm).all().item())
        if not ok:
            self.skipped += 1
            self.good = 0
            if self.log2 > 0:
                self.log2 -= 1
                self._apply()
            return False
        self.good += self.check_every
        if self.good >= self.growth_interval and self.log2 < self.max_log2:
            self.log2 += 1
            self.good = 0
            self._apply()
        return True
