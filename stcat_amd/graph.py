"""Whole-step hipGraph: forward + loss + backward captured once, replayed with one launch.

The hot path issues ~3 500 kernel launches per step, ~2 000 of them through the C ABI from Python autograd
functions; measured host enqueue time is ~77 ms of an ~88 ms step at C3 (bench.py `host_enqueue_ms_per_step`), so
the GPU-side gains of faster kernels are capped by the launch rate.  A captured step removes the host from the
loop entirely.  What makes the step capturable:

* no host synchronisation inside it (LossPlan is built from the targets beforehand; the loss's box-count
  all-reduce lives in the plan; PostProcess is outside the training step);
* every buffer the kernels see is static: activations / gradients come from the graph's private memory pool,
  weight-gradient accumulators from the zero arena (bump allocator restarted — and re-zeroed — by the captured
  step itself);
* dropout counters advance through a DEVICE word (ops.dropout_begin_step), so replays draw fresh masks although
  their launch arguments are frozen;
* collectives stay OUTSIDE the graph: GradBucketReducer runs in deferred mode and reduces the static gradient
  tensors after each replay (the all-reduce no longer overlaps the backbone backward; on 8 GPUs that exposes the
  ~0.3 GB exchange, a few ms, against the ~10 % the graph saves).
"""
from typing import Callable

import torch


class GraphedStep:
    def __init__(self, step_fn: Callable[[], torch.Tensor], device, warmup: int = 2):
        """step_fn: the compute part of one step (no collectives, no host syncs); returns the loss tensor.
        Runs `warmup` eager iterations on a side stream (allocator / lazy-init warm-up), then captures one."""
        device = torch.device(device)
        # Round 3 replaced this path by launch plans (stcat_amd/plans.py); its newer multi-stream schedules — the two
        # forward chains of the backbone, the deferred weight gradients of the grounding model — record allocator
        # stream uses that a capture cannot carry, so a captured step runs the round-2 schedule.
        # The two globals are scoped to warm-up + capture and restored afterwards (they used to stay set for
        # the rest of the process, so every later eager step silently ran the round-2 schedule).
        from . import backbone, composite
        saved = (backbone.FORWARD_CHAINS, composite.DEFER_WGRADS)
        backbone.FORWARD_CHAINS = 1
        composite.DEFER_WGRADS = False
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    step_fn()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.output = step_fn()
        finally:
            backbone.FORWARD_CHAINS, composite.DEFER_WGRADS = saved

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.output


class GraphedDecoder:
    """Decoder-scope hipGraph (BASELINE.json configs[4]; STCATNet.capture_decoder): `fn(*tensors) -> tuple of tensors`
    — the template generator, the box decoder, the time decoder on its forked stream and the heads under no_grad —
    captured once per input signature (shapes, dtypes, strides) and replayed with one hipGraphLaunch.

    * inputs are copied into static buffers before a replay (memory + positions: 2 x [T,S',256] fp32, 50 MB at T = 128;
      the masks and the two [CLS] tensors are a few KB); outputs are the graph's static tensors — valid until the next
      call with the same signature, like a launch plan's;
    * the capture follows two eager warm-up calls on a side stream (lazy caches: transposed weights, sine tables, the
      measured stream placement of ops._pick_streams) — only launches are captured, never a cache fill;
    * the step's zero arena (ops.enable_zero_arena) is out of reach during capture: an accumulator taken from it would
      be zero when captured and stale at every replay, so the skinny Linear launches take their plain (non-accumulating)
      form inside the graph;
    * the forked time decoder is part of the graph: its stream is joined into the capturing stream before the capture
      ends (ops.fork_stream)."""

    MAX_GRAPHS = 4

    def __init__(self, fn: Callable):
        self.fn = fn
        self.graphs = {}
        self.replays = 0

    @staticmethod
    def _sig(args):
        return tuple((tuple(a.shape), a.dtype, tuple(a.stride()), str(a.device)) if torch.is_tensor(a) else a for a in args)

    def _capture(self, args):
        from . import ops
        dev = next(a.device for a in args if torch.is_tensor(a))
        static_in = [a.clone() if torch.is_tensor(a) else a for a in args]
        arena = ops._ARENA.pop(str(dev), None)
        try:
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):
                    self.fn(*static_in)
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g):
                static_out = self.fn(*static_in)
        finally:
            if arena is not None:
                ops._ARENA[str(dev)] = arena
        return g, static_in, static_out

    def __call__(self, *args):
        assert not torch.is_grad_enabled(), "the captured decoder is the inference path; training runs on launch plans"
        key = self._sig(args)
        got = self.graphs.get(key)
        if got is None:
            if len(self.graphs) >= self.MAX_GRAPHS:
                self.graphs.pop(next(iter(self.graphs)))
            got = self.graphs[key] = self._capture(args)
        g, static_in, static_out = got
        for s, a in zip(static_in, args):
            if torch.is_tensor(s) and s.data_ptr() != a.data_ptr():
                s.copy_(a)
        g.replay()
        self.replays += 1
        return static_out
