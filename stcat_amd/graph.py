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
        from . import backbone, composite
        backbone.FORWARD_CHAINS = 1
        composite.DEFER_WGRADS = False
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

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.output
