#!/usr/bin/env python3
"""Where the wall time of a step goes, from a rocprofv3 --kernel-trace CSV (start/end timestamp per dispatch).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python bench.py --steps 3 --warmup 1 ...
    python tools/timeline.py /tmp/tl [--steps 3]

Splits the device timeline of the LAST traced step into classes (heavy GEMMs = dispatches >= 40 us, small = the
rest) and reports: busy time (union of intervals), idle gaps, wall time during which ONLY small kernels run (the
latency-bound tail), and the launch count — the quantities VERDICT r01 #8 asks for."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    # steps are delimited by the frame-stack stem kernel (one per forward)
    stems = [i for i, r in enumerate(rows) if "igemm_stem_kernel" in r[2] or "stem_pl_kernel" in r[2]]
    if len(stems) < 2:
        print("need >= 2 steps in the trace")
        return
    a, b = stems[-2], stems[-1]
    step = rows[a:b]
    t0, t1 = step[0][0], rows[b][0]
    wall = (t1 - t0) / 1e6
    HEAVY = 40_000  # ns
    ev = []
    step_q = step
    step = [(s, e, n) for s, e, n, _ in step]
    for s, e, n in step:
        heavy = (e - s) >= HEAVY
        ev.append((s, 1, heavy))
        ev.append((e, -1, heavy))
    ev.sort()
    busy = only_small = idle = 0
    nh = ns = 0
    prev = t0
    for t, dlt, heavy in ev:
        span = t - prev
        if span > 0:
            if nh + ns == 0:
                idle += span
            else:
                busy += span
                if nh == 0:
                    only_small += span
        prev = t
        if heavy:
            nh += dlt
        else:
            ns += dlt
    idle += max(0, t1 - prev)
    small = [(e - s) for s, e, n in step if (e - s) < HEAVY]
    heavy = [(e - s) for s, e, n in step if (e - s) >= HEAVY]
    print(f"step wall {wall:.2f} ms | launches {len(step)} (heavy {len(heavy)}, small {len(small)})")
    print(f"  busy {busy/1e6:.2f} ms, idle gaps {idle/1e6:.2f} ms, only-small-kernels wall {only_small/1e6:.2f} ms")
    print(f"  sum heavy {sum(heavy)/1e6:.2f} ms, sum small {sum(small)/1e6:.2f} ms (avg small {sum(small)/max(1,len(small))/1e3:.1f} us)")
    # phases: forward backbone ends at the first non-conv heavy...: report the wall time between landmark kernels
    marks = {"first mha_self_fwd": None, "first attn_q1_fwd": None, "first attn_q1_bwd": None, "last mha_self_bwd": None,
             "first igemm_pl_wgrad": None}
    for s, e, n in step:
        if "mha_self_fwd" in n and marks["first mha_self_fwd"] is None:
            marks["first mha_self_fwd"] = s
        if "attn_q1_fwd" in n and marks["first attn_q1_fwd"] is None:
            marks["first attn_q1_fwd"] = s
        if "attn_q1_bwd" in n and marks["first attn_q1_bwd"] is None:
            marks["first attn_q1_bwd"] = s
        if "mha_self_bwd" in n:
            marks["last mha_self_bwd"] = e
        if ("igemm_pl_wgrad" in n or "igemm_bs_wgrad_kernel<256" in n) and marks["first igemm_pl_wgrad"] is None:
            marks["first igemm_pl_wgrad"] = s
    # (round 5: the encoder FFN's weight gradients run on the plane weight-gradient kernel too, INSIDE the grounding section —
    # the section ends at the first plane weight gradient BEHIND the last self-attention backward: the backbone's)
    last_mha = marks["last mha_self_bwd"]
    bb = next((s for s, e, n in step if "igemm_pl_wgrad" in n and last_mha is not None and s >= last_mha), None)
    marks["first backbone igemm_pl_wgrad"] = bb
    print("  landmarks (ms from step start): " + ", ".join(f"{k} {((v - t0)/1e6):.2f}" for k, v in marks.items() if v))
    if marks["first mha_self_fwd"] and bb:
        print(f"  grounding window (first mha_self_fwd .. first backbone weight gradient): {(bb - marks['first mha_self_fwd'])/1e6:.2f} ms"
              " (traced: the tracer serialises the streams' small launches; tools/node_times.py gives the untraced figure)")
    agg = {}
    for s, e, n in step:
        k = n.split("(")[0][:60]
        a_ = agg.setdefault(k, [0, 0])
        a_[0] += 1
        a_[1] += e - s
    for k, (c, tns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"    {tns/1e6:7.3f} ms {c:5d}x  {k}")
    # hardware queues: HIP maps its streams onto GPU_MAX_HW_QUEUES (default 4) HSA queues; streams that share one serialise
    qs = {}
    for s, e, n, q in step_q:
        a_ = qs.setdefault(q, [0, 0, s, e, {}])
        a_[0] += 1
        a_[1] += e - s
        a_[2] = min(a_[2], s)
        a_[3] = max(a_[3], e)
        k = n.split("(")[0].replace("void ", "")[:28]
        a_[4][k] = a_[4].get(k, 0) + (e - s)
    print(f"  hardware queues used in the step: {len(qs)}")
    for q, (c, tns, s0, e0, names) in sorted(qs.items(), key=lambda kv: -kv[1][1]):
        top = ", ".join(f"{k} {v/1e6:.1f}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4])
        print(f"    queue {q}: {c:5d} launches, {tns/1e6:7.2f} ms of kernels, active {((s0 - t0)/1e6):6.2f} .. {((e0 - t0)/1e6):6.2f} ms | {top}")
    # library (at::native / rocPRIM) kernels by functor: what is still not ours
    import re
    lib = {}
    for s, e, n in step:
        if "at::native" in n or "rocprim" in n or "hipcub" in n:
            m = re.search(r"(\w*Functor\w*|direct_copy_kernel\w*|CatArrayBatchedCopy\w*|reduce_kernel|\w*fill\w*|index\w*kernel\w*)", n)
            k = (n.split("<")[0].split("::")[-1] + " / " + (m.group(1) if m else n[40:120]))
            a_ = lib.setdefault(k, [0, 0])
            a_[0] += 1
            a_[1] += e - s
    print(f"  library kernels: {sum(c for c, _ in lib.values())} launches, {sum(t for _, t in lib.values())/1e6:.3f} ms")
    for k, (c, tns) in sorted(lib.items(), key=lambda kv: -kv[1][0])[:20]:
        print(f"    {c:5d}x {tns/1e6:7.3f} ms  {k}")


if __name__ == "__main__":
    main()
