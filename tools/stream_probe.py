#!/usr/bin/env python3
"""Which HIP streams really run beside each other?  Raw timings of the probe behind stcat_amd.ops._pick_streams:
a 300 us one-workgroup spin kernel on two streams at once (concurrent: ~0.3 ms, sharing a hardware queue: ~0.6 ms).
    python tools/stream_probe.py [--comm]      (--comm: with a live 1-rank RCCL process group, as STCAT_FORCE_COMM=1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stcat_amd import _lib as L  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if "--comm" in sys.argv:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        from stcat_amd.dist import init_rccl_process_group
        init_rccl_process_group(dev)
        x = torch.ones(1 << 20, device=dev)
        dist.all_reduce(x)
        torch.cuda.synchronize()
    lib = L.load()
    main_s = torch.cuda.current_stream(dev)
    cands = [torch.cuda.Stream(device=dev) for _ in range(12)]
    us = 300

    def run(streams):
        streams = [s_ for s_ in streams if s_ is not main_s] + [s_ for s_ in streams if s_ is main_s]
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for st in streams:
            if st is not main_s:
                st.wait_event(e0)
        for st in streams:
            assert lib.stcat_spin(us, st.cuda_stream) == 0
        for st in streams:
            if st is not main_s:
                main_s.wait_stream(st)
        e1.record(main_s)
        e1.synchronize()
        return e0.elapsed_time(e1)

    for _ in range(2):
        run([main_s]); run([main_s, cands[0]])
    print("single on main        :", " ".join(f"{run([main_s]):.3f}" for _ in range(4)))
    print("single on cand 0      :", " ".join(f"{run([cands[0]]):.3f}" for _ in range(4)))
    print("main + cand i         :", " ".join(f"{run([main_s, c]):.3f}" for c in cands))
    print("cand 0 + cand i       :", " ".join(f"{run([cands[0], c]):.3f}" for c in cands[1:]))
    print("cand 1 + cand i       :", " ".join(f"{run([cands[1], c]):.3f}" for c in cands[2:]))
    print("main + c0 + c1 + c2   :", f"{run([main_s, cands[0], cands[1], cands[2]]):.3f}")
    print("handles:", [hex(c.cuda_stream) for c in cands[:6]])


if __name__ == "__main__":
    main()
