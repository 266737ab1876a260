"""Launch plans: the host side of a composite node as ONE C call (csrc/launch_plan.h, `stcat_plan_*`).

Round 2 issued each of a step's ~1700 launches as its own Python -> ctypes call: 29 ms of host work per step at any
clip size, and the driver's run was host-bound.  The composite nodes (backbone, encoder, box / time
decoder, heads: the call order of models/pipeline.py:52-121) are straight launch sequences whose arguments depend on the
input SHAPES only, so each is recorded once and replayed:

* first call of a (node, input signature): plain eager Python (fills the per-module caches);
* second call: the same Python path runs under a `Recorder` — every `_lib.call` is also appended to a C++ plan, every
  tensor it allocates comes from the plan's private `torch.cuda.MemPool` (kept for the life of the plan, so the baked
  addresses stay valid), zero-initialised accumulators come from plan-owned chunks, each cleared by ONE memset op at
  the point of the sequence where the chunk was opened;
* from the third call on: `stcat_plan_run` — pointers into the tensors the caller passes in per step (inputs,
  parameters, upstream gradients: the "externals") are patched by a relocation list, everything else is static.

Semantics are those of a captured graph with static buffers: a node's outputs and parameter gradients of step N live in
the same memory as those of step N-1 (the training loop must reset `.grad` to None between steps — PyTorch's default
`zero_grad(set_to_none=True)`; this is checked), and a second forward of the same signature while the first one's
backward is still outstanding falls back to the eager path.  Plans are opt-in: `plans.enable()`.
"""
from __future__ import annotations

import bisect
import ctypes
import os
import struct
import time
import weakref
from typing import Dict, List, Optional

import torch
from torch.autograd import Function

from . import _lib as L

ENABLED = bool(os.environ.get("STCAT_PLANS"))
STRICT = bool(os.environ.get("STCAT_PLAN_STRICT"))   # raise when a recorded region runs a kernel that is not ours
AUDIT = bool(os.environ.get("STCAT_PLAN_AUDIT"))      # debug: Recorder.finalize lists baked pointers it cannot account for
AUDIT_LOG = []
KEEPALL = bool(os.environ.get("STCAT_PLAN_KEEPALL"))  # debug: no block of the plan's pool is reused inside a recording
WARMUP_CALLS = 1
STATIC_EPOCH = 0     # bumped whenever a cached device object the plans point at is rebuilt (FrozenBN fold, weight planes)
MAX_PLANS_PER_NODE = 8
_M64 = (1 << 64) - 1
STATS = {"recorded": 0, "replayed": 0, "eager": 0, "run_s": 0.0}


def enable(on: bool = True) -> None:
    global ENABLED
    ENABLED = bool(on)


def invalidate() -> None:
    """forget every plan's validity (a cached static tensor moved: load_state_dict, .to(), a rebuilt weight table)"""
    global STATIC_EPOCH
    STATIC_EPOCH += 1


def clear() -> None:
    """drop every plan and its memory pool"""
    for cache in _CACHES:
        cache.clear()
    invalidate()
    from . import ops
    ops.free_wgrad_workspaces()      # (their addresses were baked into the plans dropped above)
    # the dropped entries own torch.cuda.MemPool objects and sit in reference cycles (ctx <-> tensors): collect them NOW.
    # Left to the cyclic GC, a pool's destructor may run in the middle of a later recording — inside
    # torch.cuda.use_mem_pool — where the caching allocator aborts the process (captures_underway.empty() assert; seen in
    # round 5 with bench.py's mode loop under a live RCCL group)
    import gc
    gc.collect()


_CACHES: List[list] = []


class InvalidatesPlans:
    """nn.Module mixin of the seam modules (vision encoder, cross-modal encoder, query decoder, STCATNet): whatever moves
    or rewrites their parameters / buffers behind a plan's back — `.to()` / `.cuda()` / `.double()` (nn.Module._apply),
    `load_state_dict` (in-place copies: same pointers, new FrozenBN statistics) — bumps STATIC_EPOCH, so no recorded
    plan matches any more and the next steps run eager -> record -> replay again (nothing ever called
    invalidate(); replays skip FrozenBatchNorm2d.folded() and the weight-table key checks)."""

    def _apply(self, fn, *a, **k):
        invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        invalidate()
        return super().load_state_dict(*a, **k)


def _extent_bytes(t: torch.Tensor) -> int:
    if t.numel() == 0:
        return 0
    last = 0
    for s, st in zip(t.shape, t.stride()):
        last += (s - 1) * st
    return (last + 1) * t.element_size()


_FBITS = struct.Struct("<f")
_UBITS = struct.Struct("<I")

# aten ops that launch nothing (views / allocation): anything else inside a recorded region would not be replayed
_VIEW_OPS = ("empty", "view", "as_strided", "detach", "alias", "slice", "select", "expand", "permute", "reshape",
             "transpose", "unsqueeze", "squeeze", "_unsafe_view", "unbind", "split", "narrow", "new_empty",
             "_reshape_alias", "lift_fresh", "unfold", "is_pinned", "record_stream", "chunk", "sym_", "stride", "size",
             "numel", "is_contiguous", "storage_offset", "dim", "t")


class _Watch(torch.utils._python_dispatch.TorchDispatchMode):
    """keeps every tensor an aten op returns alive (CPU emulator backend: no memory pool) and notes the ops that are not
    views / allocations — those run at record time only, a replay would silently skip them"""

    def __init__(self, rec, keepalive: bool):
        super().__init__()
        self.rec = rec
        self.keepalive = keepalive

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if self.keepalive:
            if torch.is_tensor(out):
                self.rec.keep.append(out)
            elif isinstance(out, (tuple, list)):
                self.rec.keep.extend(o for o in out if torch.is_tensor(o))
        if not self.rec.allow_foreign:
            name = str(func)                               # "aten.add_.Tensor"
            base = name.split(".")[1] if "." in name else name
            if not any(base.startswith(v) for v in _VIEW_OPS):
                # an out-of-place op whose result aliases its input did nothing (x.contiguous() / .to() of a fitting x)
                alias = (torch.is_tensor(out) and args and torch.is_tensor(args[0]) and not base.endswith("_")
                         and out.data_ptr() == args[0].data_ptr())
                if not alias:
                    self.rec.foreign.append(name)
        return out


class Recorder:
    """Installed as `_lib.RECORDER` while a node's Python path runs: mirrors every C-ABI launch into a C++ plan."""

    def __init__(self, device: torch.device, externals: List[torch.Tensor]):
        self.lib = L.load()
        self.h = self.lib.stcat_plan_create()
        self.dev = device
        self.cuda = device.type == "cuda"
        cur = L.stream_of(torch.empty(0, device=device)) if self.cuda else None
        self.slots = {cur: 0}
        self.side_handles: List[int] = []       # raw hipStream_t of slots 1..
        self.side_streams: List[object] = []    # the torch.cuda.Stream objects (keepalive)
        self.calls = []                         # (first word, signature, words)
        self.call_names = []                    # (STCAT_PLAN_AUDIT)
        self.memsets = []
        self.effects = []
        self.prereqs = []
        self.yields = []
        self.keep: List[torch.Tensor] = []
        self.foreign: List[str] = []
        self.allow_foreign = False
        self.zero_chunks = {}                   # stream slot -> [[buffer, used elements, memset word], ...]
        self.ext = externals
        self._fn = {}
        from . import ops
        self.drop0 = ops._dropout_stream.offset
        self.n_calls = 0

    # ---- called from _lib.call ------------------------------------------------------------------------------------
    def add_call(self, name: str, args) -> None:
        sig = L.SIGNATURES[name]
        if not sig.endswith("s") or "P" in sig:
            if name.startswith("stcat_plan_"):
                return
            raise L.StcatHipError(f"{name} cannot be part of a launch plan (not a stream-ordered launch)")
        fn = self._fn.get(name)
        if fn is None:
            fn = self._fn[name] = self.lib.stcat_plan_fn_index(name.encode())
            if fn < 0:
                raise L.StcatHipError(f"{name} is not in the plan table of the library")
        words, slot = [], 0
        for k, a in zip(sig, args):
            if k == "p":
                words.append(int(a) if a else 0)
            elif k == "s":
                words.append(0)
                slot = self._slot(a)
            elif k == "f":
                words.append(_UBITS.unpack(_FBITS.pack(float(a)))[0])
            else:
                words.append(int(a) & _M64)
        n = len(words)
        w0 = self.lib.stcat_plan_add_call(self.h, fn, (ctypes.c_ulonglong * n)(*words), n, slot, n - 1)
        if w0 < 0:
            raise L.StcatHipError(f"plan_add_call({name}): {self.lib.stcat_last_error().decode()}")
        self.calls.append((w0, sig, words))
        if AUDIT:
            self.call_names.append(name)
        self.n_calls += 1

    def _slot(self, raw, stream_obj=None) -> int:
        s = self.slots.get(raw)
        if s is None:
            s = self.slots[raw] = len(self.slots)
            self.side_handles.append(raw)
            self.side_streams.append(stream_obj)
            if s > 250:
                raise L.StcatHipError("launch plan: too many streams")
        return s

    # ---- called from the stream helpers of stcat_amd.ops ----------------------------------------------------------
    def wait(self, waiter, signal) -> None:
        """`waiter.wait_stream(signal)` (torch.cuda.Stream objects) as a plan op"""
        a = self._slot(waiter.cuda_stream, waiter)
        b = self._slot(signal.cuda_stream, signal)
        if self.lib.stcat_plan_add_wait(self.h, a, b) != 0:
            raise L.StcatHipError(self.lib.stcat_last_error().decode())

    def host_call(self, fn):
        """a host-side action in the middle of the sequence (the DP reducer's early bucket delivery): runs now and at
        this point of every replay, under the stream that is current now"""
        stream = torch.cuda.current_stream(self.dev) if self.cuda else None
        if stream is not None:
            self._slot(stream.cuda_stream, stream)
        self.yields.append((fn, stream))
        if self.lib.stcat_plan_add_yield(self.h, len(self.yields) - 1) != 0:
            raise L.StcatHipError(self.lib.stcat_last_error().decode())
        # The action is host code by definition — bucket copies, the asynchronous all-reduce of a completed bucket
        # (c10d.allreduce_) — and runs again at this point of every replay: the dispatch watch must not count its kernels
        # as "foreign kernels of the node body".  (Round 6: it did, so with a live process group the backbone's backward
        # recording was REFUSED and the whole backbone node stayed on the eager path: +6 ms of host enqueue each way and
        # +3.4 ms per step at one rank, profiles/r06_prefix_pipeline.log — every N > 1 run since the watch became
        # unconditional in round 4 carried that.)
        prev, self.allow_foreign = self.allow_foreign, True
        try:
            return fn()
        finally:
            self.allow_foreign = prev

    def effect(self, fn) -> None:
        """a host-side state change of the region (dropout bookkeeping): repeated after every replay"""
        self.effects.append(fn)

    def prereq(self, fn) -> None:
        """a host-side check the region relied on WITHOUT launching anything (a cached W^T that was current at record
        time, ops.LinearTransposes.get): run before every replay, on the replay's stream; it may launch eagerly"""
        self.prereqs.append(fn)

    def zeros(self, shape) -> torch.Tensor:
        """a zeroed fp32 buffer from the plan's accumulation chunks.  A chunk is cleared by ONE memset op that sits at the
        point of the sequence where the chunk was opened, on the stream that opened it (its block may have served an
        earlier temporary of the same recording: clearing it at the top of the plan would be undone by that temporary's
        kernels), and chunks are per stream (the weight-gradient stream's accumulators are cleared on that stream)."""
        n = 1
        for d in shape:
            n *= int(d)
        stream = torch.cuda.current_stream(self.dev) if self.cuda else None
        slot = self._slot(stream.cuda_stream, stream) if stream is not None else 0
        chunks = self.zero_chunks.setdefault(slot, [])
        if chunks:
            ch = chunks[-1]
            start = (ch[1] + 63) & ~63
            if start + n <= ch[0].numel():
                ch[1] = start + n
                return ch[0][start:start + n].view(*shape)
        size = max(n, min(1 << 24, (1 << 20) << len(chunks)))
        self.allow_foreign = True
        try:
            buf = torch.zeros(size, device=self.dev, dtype=torch.float32)
        finally:
            self.allow_foreign = False
        self.keep.append(buf)
        w = self.lib.stcat_plan_add_memset(self.h, buf.data_ptr(), size * 4, slot, 0)
        if w < 0:
            raise L.StcatHipError(self.lib.stcat_last_error().decode())
        chunks.append([buf, n, w])
        return buf[:n].view(*shape)

    # ---- finish ---------------------------------------------------------------------------------------------------
    def finalize(self) -> "Plan":
        lib = self.lib
        ranges = sorted((t.data_ptr(), t.data_ptr() + _extent_bytes(t), i) for i, t in enumerate(self.ext)
                        if t is not None and t.numel() > 0)
        los = [r[0] for r in ranges]
        # widest-first: overlapping externals (views of one buffer) resolve to the first, which moves with the others
        n_reloc = 0
        for w0, sig, words in self.calls:
            for j, k in enumerate(sig):
                if k != "p" or words[j] == 0:
                    continue
                p = words[j]
                i = bisect.bisect_right(los, p) - 1
                while i >= 0:
                    lo, hi, e = ranges[i]
                    if lo <= p < hi:
                        if lib.stcat_plan_add_reloc(self.h, w0 + j, e, p - lo) != 0:
                            raise L.StcatHipError(lib.stcat_last_error().decode())
                        n_reloc += 1
                        break
                    if p - lo > (1 << 36):
                        break
                    i -= 1
        if AUDIT:
            # debug aid (STCAT_PLAN_AUDIT=1, emulator backend: every tensor of the recording is in self.keep): list the
            # baked pointers that are neither externals nor tensors of this recording — module-level caches are expected
            # here (weight planes, FrozenBN folds, transposes, workspaces); anything else dangles after the step
            own = sorted((t.data_ptr(), t.data_ptr() + _extent_bytes(t)) for t in self.keep if t.numel() > 0)
            olos = [r[0] for r in own]
            seen = {}
            for (w0, sig, words), name in zip(self.calls, self.call_names):
                for j, k in enumerate(sig):
                    if k != "p" or words[j] == 0:
                        continue
                    p = words[j]
                    i = bisect.bisect_right(los, p) - 1
                    if i >= 0 and ranges[i][0] <= p < ranges[i][1]:
                        continue
                    i = bisect.bisect_right(olos, p) - 1
                    if i >= 0 and own[i][0] <= p < own[i][1]:
                        continue
                    seen.setdefault((name, j), set()).add(p)
            AUDIT_LOG.append({k: sorted(v) for k, v in seen.items()})
        for chunks in self.zero_chunks.values():
            for buf, used, w in chunks:        # the memset clears what was handed out, not the whole chunk
                if lib.stcat_plan_set_word(self.h, w + 1, used * 4) != 0:
                    raise L.StcatHipError(lib.stcat_last_error().decode())
        from . import ops
        return Plan(self, ops._dropout_stream.offset, n_reloc)


class Plan:
    def __init__(self, rec: Recorder, drop1: int, n_reloc: int):
        self.lib = rec.lib
        self.h = rec.h
        self.dev = rec.dev
        self.cuda = rec.cuda
        self.side_handles = list(rec.side_handles)
        self.side_streams = list(rec.side_streams)
        self.effects = list(rec.effects)
        self.prereqs = list(rec.prereqs)
        self.yields = list(rec.yields)
        self.keep = rec.keep
        self.n_ext = len(rec.ext)
        self.drop = (rec.drop0, drop1)
        self.n_calls = rec.n_calls
        self.n_reloc = n_reloc
        self.foreign = list(rec.foreign)
        ns = 1 + len(self.side_handles)
        self._streams = (ctypes.c_void_p * ns)(None, *self.side_handles)
        self._next = ctypes.c_int(0)
        self._tag = ctypes.c_int(0)
        self._ext_t = ctypes.c_ulonglong * max(self.n_ext, 1)
        self._fin = weakref.finalize(self, self.lib.stcat_plan_destroy, self.h)

    def run(self, ext: List[Optional[torch.Tensor]]) -> None:
        lib = self.lib
        t0 = time.perf_counter()
        for f in self.prereqs:
            f()
        extv = self._ext_t(*[0 if t is None else t.data_ptr() for t in ext])
        if self.cuda:
            self._streams[0] = torch._C._cuda_getCurrentRawStream(self.dev.index if self.dev.index is not None
                                                                  else torch.cuda.current_device())
        start = 0
        while True:
            rc = lib.stcat_plan_run(self.h, extv, self.n_ext, self._streams, len(self._streams), start,
                                    ctypes.byref(self._next), ctypes.byref(self._tag))
            if rc != 0:
                raise L.StcatHipError(f"stcat_plan_run failed (rc={rc}): {lib.stcat_last_error().decode()}")
            if self._next.value < 0:
                break
            fn, stream = self.yields[self._tag.value]
            if stream is not None and stream.cuda_stream != self._streams[0]:
                with torch.cuda.stream(stream):
                    fn()
            else:
                fn()
            start = self._next.value
        if self.drop[1] != self.drop[0]:
            from . import ops
            ops._dropout_stream.offset = self.drop[1]
        for e in self.effects:
            e()
        STATS["replayed"] += 1
        STATS["run_s"] += time.perf_counter() - t0


class _ShimCtx:
    """the part of the autograd ctx API the node bodies use; lives as long as its plan (its tensors are static)"""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()
        self.materialize = True
        self.static = False      # True: recorded into a plan — the node must not drop its state in backward

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, v):
        self.materialize = bool(v)


class _Entry:
    __slots__ = ("calls", "fwd", "bwd", "ctx", "outs", "pool", "live", "done", "single", "params", "grad_ptrs", "gsig",
                 "spec", "param_pos", "refused", "param_ptrs")

    def __init__(self):
        self.calls = 0
        self.fwd = None          # Plan
        self.bwd = {}            # grad-output signature -> (Plan, grads)
        self.ctx = None          # the node's own ctx object of the recording pass (static tensors)
        self.outs = None
        self.pool = None
        self.live = None         # weakref to the autograd node of the forward whose backward is outstanding
        self.done = True
        self.single = False
        self.params = None
        self.param_pos = None
        self.grad_ptrs = set()
        self.param_ptrs = ()
        self.refused = False     # a recording of this node ran kernels that are not ours: stays eager


def _is_param(t) -> bool:
    return t.requires_grad and t.is_leaf


_TENSOR = object()


def _spec_of(args):
    """what a later call must present at each argument position to take the same plan: the same object (parameters,
    modules, None), an equal python value, or a tensor of the same geometry"""
    spec = []
    for a in args:
        if torch.is_tensor(a) and not _is_param(a):
            spec.append((_TENSOR, a.shape, a.stride(), a.dtype, a.requires_grad, a.data_ptr() % 16))
        else:
            spec.append(a)
    return spec


def _matches(spec, args) -> bool:
    if len(spec) != len(args):
        return False
    for s, a in zip(spec, args):
        if s is a:
            continue
        if type(s) is tuple and len(s) == 6 and s[0] is _TENSOR:
            if not (torch.is_tensor(a) and a.shape == s[1] and a.stride() == s[2] and a.dtype is s[3]
                    and a.requires_grad == s[4] and a.data_ptr() % 16 == s[5]) or _is_param(a):
                return False
        elif torch.is_tensor(s) or torch.is_tensor(a) or s != a:
            return False
    return True


def _global_sig(dev):
    from . import ops
    ds = ops._dropout_stream
    base = ds.base(dev).data_ptr() if dev in ds._base else 0
    return (L.get_mma_mode(), STATIC_EPOCH, ds.offset, ds.seed, base, id(ops.GRAD_SINK), ops.FORK_ENABLED,
            ops.WGRAD_STREAM_ENABLED)


def _record(dev, ext, pool, body):
    """run body() with every launch mirrored into a new plan; returns (result, Plan)"""
    rec = Recorder(dev, ext)
    cuda = dev.type == "cuda"
    # The dispatch watch is on for EVERY recording (one pass per plan): an aten kernel inside the node body (a
    # .contiguous() of a strided input, an expand-sum) would run now and be missing from every replay
    watch = _Watch(rec, keepalive=(not cuda) or KEEPALL)
    prev = L.RECORDER
    L.RECORDER = rec
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()          # (no destructor of an old plan's memory pool inside use_mem_pool: see clear())
    try:
        if cuda:
            with torch.cuda.use_mem_pool(pool, device=dev):
                if watch is not None:
                    with watch:
                        res = body()
                else:
                    res = body()
        else:
            with watch:
                res = body()
    finally:
        L.RECORDER = prev
        if gc_was_on:
            gc.enable()
    plan = rec.finalize()
    if rec.foreign:
        if STRICT:
            raise L.StcatHipError(f"launch plan: the recorded region ran kernels that are not ours: {sorted(set(rec.foreign))}")
        # not replayable: the caller keeps the results of this pass (every launch did execute) and stays eager
        STATS["refused"] = STATS.get("refused", 0) + 1
        _warn_foreign(sorted(set(rec.foreign)))
        return res, None
    STATS["recorded"] += 1
    return res, plan


_WARNED = set()


def _warn_foreign(names) -> None:
    key = tuple(names)
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(f"stcat_amd launch plan refused (the node body ran kernels that are not ours: {list(names)}); "
                      "this node stays on the eager path")


class PlannedFn(Function):
    """`PlannedFn.apply(NodeFn, *args)` == `NodeFn.apply(*args)` for a node whose forward / backward are launch sequences
    (stcat_amd.composite, stcat_amd.backbone): eager the first time a signature is seen, recorded the second, replayed
    afterwards."""

    @staticmethod
    def forward(ctx, node, *args):
        nig = tuple(ctx.needs_input_grad[1:])
        dev = next(a.device for a in args if torch.is_tensor(a))
        cache = node.__dict__.get("_plan_cache")
        if cache is None:
            cache = []
            node._plan_cache = cache
            _CACHES.append(cache)
        gsig = (nig, _global_sig(dev))
        e = None
        for i, cand in enumerate(cache):          # most recently used first
            if cand.gsig == gsig and _matches(cand.spec, args):
                e = cand
                if i:
                    cache.insert(0, cache.pop(i))
                break
        if e is None:
            if len(cache) >= MAX_PLANS_PER_NODE:
                cache.pop()
            e = _Entry()
            e.gsig, e.spec = gsig, _spec_of(args)
            cache.insert(0, e)
        e.calls += 1
        ctx.node = node
        ctx.entry = None
        if e.fwd is not None and e.param_ptrs != tuple(p.data_ptr() for p in e.params):
            # a parameter's storage was swapped behind the plan (p.data = ..., an optimizer that replaces tensors): the
            # per-module tables the plan points at (weight planes) hold the old address — start over
            invalidate()
            cache.remove(e)
            e = _Entry()
            e.gsig, e.spec, e.calls = None, None, 1
        busy = not e.done and e.live is not None and e.live() is not None
        if e.calls <= WARMUP_CALLS or busy or not any(nig) or e.refused:
            STATS["eager"] += 1
            c = ctx.shim = _ShimCtx(nig)          # (the node indexes needs_input_grad by ITS argument positions)
            outs = node.forward(c, *args)
            if not c.materialize:
                ctx.set_materialize_grads(False)
            return outs
        tensors = [a for a in args if torch.is_tensor(a)]
        ctx.entry = e
        ctx.ext = tensors
        ctx.args = args
        e.live = weakref.ref(ctx)
        e.done = False
        if e.fwd is None:
            if dev.type == "cuda":
                e.pool = torch.cuda.MemPool()
            c = _ShimCtx(nig)
            c.static = True
            outs, e.fwd = _record(dev, tensors, e.pool, lambda: node.forward(c, *args))
            if e.fwd is None:                     # refused (foreign kernels in the body): this pass WAS a full eager pass
                e.refused, e.done, e.live = True, True, None
                c.static = False
                ctx.entry, ctx.ext, ctx.args, ctx.shim = None, None, None, c
                if not c.materialize:
                    ctx.set_materialize_grads(False)
                return outs
            e.ctx = c
            e.single = torch.is_tensor(outs)
            outs_t = (outs,) if e.single else tuple(outs)
            e.outs = tuple(o.detach() if torch.is_tensor(o) else o for o in outs_t)
            e.params = [a for a in args if torch.is_tensor(a) and _is_param(a)]
            e.param_ptrs = tuple(p.data_ptr() for p in e.params)
            e.param_pos = [i for i, a in enumerate(args) if torch.is_tensor(a) and _is_param(a)]
        else:
            e.fwd.run(tensors)
            outs_t = tuple(o.detach() if torch.is_tensor(o) else o for o in e.outs)
            outs = outs_t[0] if e.single else outs_t
        if not e.ctx.materialize:
            ctx.set_materialize_grads(False)
        nd = [o for o in outs_t if torch.is_tensor(o) and not o.dtype.is_floating_point]
        if nd:
            ctx.mark_non_differentiable(*nd)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        e = ctx.entry
        node = ctx.node
        if e is None:
            r = node.backward(ctx.shim, *gouts)
            ctx.shim = None
            return (None,) + tuple(r)
        if e.done:
            raise L.StcatHipError("launch plan: backward ran twice for one forward (retain_graph is not supported; "
                                  "disable plans with stcat_amd.plans.enable(False))")
        gouts = tuple(g if (g is None or g.is_contiguous()) else g.contiguous() for g in gouts)
        ctx_args = ctx.args
        for p in e.params:
            if p.grad is not None and p.grad.data_ptr() in e.grad_ptrs:
                raise L.StcatHipError(
                    "launch plan: a parameter's .grad still aliases the plan's static gradient buffer — reset gradients "
                    "to None between steps (optimizer.zero_grad(set_to_none=True), PyTorch's default); gradient "
                    "accumulation over several backward passes needs stcat_amd.plans.enable(False)")
        bkey = tuple(None if g is None else (tuple(g.shape), g.dtype) for g in gouts)
        ext = list(ctx.ext) + [g for g in gouts]
        got = e.bwd.get(bkey)
        if got is None:
            dev = ext[0].device
            grads, plan = _record(dev, ext, e.pool, lambda: node.backward(e.ctx, *gouts))
            grads = tuple(grads)
            if plan is None:
                e.refused = True                  # later forwards of this signature go eager
            else:
                e.bwd[bkey] = (plan, tuple(g.detach() if torch.is_tensor(g) else g for g in grads))
                e.grad_ptrs.update(g.data_ptr() for g in grads if torch.is_tensor(g))
        else:
            plan, static = got
            plan.run(ext)
            grads = tuple(g.detach() if torch.is_tensor(g) else g for g in static)
        e.done = True
        # Data-parallel run: hand the node's parameter gradients to the reducer in ONE call, from the stream that wrote
        # them, and return None for them — no AccumulateGrad node and no per-parameter Python hook runs for ~95 % of the
        # model's parameters (round 2: ~1000 hook calls on the autograd thread per step delayed the next node's launch;
        # a live process group cost 3.7 ms per step at one rank, profiles/r03_bench_variants.log)
        from . import ops
        sink = ops.GRAD_SINK
        if sink is not None and e.param_pos:
            pos = [i for i in e.param_pos if torch.is_tensor(grads[i])]
            if pos and sink.early([ctx_args[i] for i in pos], [grads[i] for i in pos]):
                gl = list(grads)
                for i in pos:
                    gl[i] = None
                grads = tuple(gl)
        ctx.ext = None
        ctx.args = None
        return (None,) + grads


def apply(node, *args):
    """entry point used by the model code: through the plan wrapper when plans are enabled, else the node itself"""
    if ENABLED and torch.is_grad_enabled():
        return PlannedFn.apply(node, *args)
    return node.apply(*args)
