"""Input side of the hot path (scripts/train_net.py:105-108 moves every batch with `.to(device)` on the
compute stream): frames go host -> HBM on a COPY stream into one of two resident device buffers while the previous
step computes, so the step never waits for PCIe.  Works on the video decoder's uint8 [T,H,W,3] frames (normalised
inside the stem's gather, 38.5 MB per C3 clip) and on the reference's normalised fp32 [T,3,H,W] tensor (154 MB) alike.
"""
from __future__ import annotations

from typing import Iterable, Iterator

import torch


class DeviceFramePrefetcher:
    """`for frames in DeviceFramePrefetcher(host_clips, device): step(frames)`.

    host_clips yields CPU tensors of one fixed shape / dtype (pinned or not: they are staged through a pinned buffer
    when they are not).  The tensor handed out stays valid until the next-but-one clip is requested (two buffers): the
    consumer's kernels of step k are ordered before the copy of clip k + 2 into the same buffer."""

    def __init__(self, host_clips: Iterable[torch.Tensor], device: torch.device, stage=None):
        """stage: optional `Backbone.stage_next` (round 6) — every time a clip is handed out, the NEXT one (already
        being copied) is declared to the backbone with its copy event, so its frozen prefix (stem + max-pool + layer1)
        runs under the current step's grounding section"""
        self.declare = stage
        self.it: Iterator[torch.Tensor] = iter(host_clips)
        self.dev = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.bufs = [None, None]
        self.stage = [None, None]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]     # copy of buffer i finished
        self.free = [None, None]                                   # consumer is done with buffer i
        self.k = 0
        self._pending = None
        self._issue()

    def _issue(self) -> None:
        try:
            clip = next(self.it)
        except StopIteration:
            self._pending = None
            return
        i = self.k & 1
        if self.bufs[i] is None or self.bufs[i].shape != clip.shape or self.bufs[i].dtype != clip.dtype:
            self.bufs[i] = torch.empty(clip.shape, dtype=clip.dtype, device=self.dev)
            # allocated on the CURRENT stream, first written on the copy stream: the block may have served kernels still
            # queued on the current stream, so the copy stream waits for them, and the allocator is told
            # about the second stream
            self.copy_stream.wait_stream(torch.cuda.current_stream(self.dev))
            self.bufs[i].record_stream(self.copy_stream)
        src = clip
        if not clip.is_pinned():
            if self.stage[i] is None or self.stage[i].shape != clip.shape or self.stage[i].dtype != clip.dtype:
                self.stage[i] = torch.empty(clip.shape, dtype=clip.dtype).pin_memory()
            self.ready[i].synchronize()            # the previous copy out of this staging buffer has left the host
            self.stage[i].copy_(clip)
            src = self.stage[i]
        with torch.cuda.stream(self.copy_stream):
            if self.free[i] is not None:
                self.copy_stream.wait_event(self.free[i])          # step k - 2 no longer reads this buffer
            self.bufs[i].copy_(src, non_blocking=True)
            self.ready[i].record(self.copy_stream)
        self._pending = i
        self.k += 1

    def __iter__(self):
        return self

    def __next__(self) -> torch.Tensor:
        if self._pending is None:
            raise StopIteration
        i = self._pending
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self.ready[i])               # device-side wait: the host does not block
        out = self.bufs[i]
        j = i ^ 1                                   # the consumer has moved on from the other buffer: mark it free
        ev = torch.cuda.Event()
        ev.record(cur)
        self.free[j] = ev
        self._issue()                               # clip k + 1 starts copying while the caller computes on clip k
        if self.declare is not None and self._pending is not None:
            self.declare(self.bufs[self._pending], self.ready[self._pending])
        return out
