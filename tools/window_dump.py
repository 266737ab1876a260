#!/usr/bin/env python3
"""Dispatch-by-dispatch listing of one window of a traced step (rocprofv3 --kernel-trace CSV).

    python tools/window_dump.py /tmp/tl [--from mha_self_fwd] [--to igemm_pl_wgrad] [--all]

Default window: the grounding section of the LAST traced step (first self-attention forward .. first plane weight
gradient).  One line per dispatch: start (us from the window start), duration, gap to the previous dispatch of the same
hardware queue, queue, grid / workgroup size, kernel.  Followed by per-kernel sums of the window and the per-queue busy
time — what VERDICT r04 item 1 asks to be cut."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]


def main():
    d = sys.argv[1]
    a_from = sys.argv[sys.argv.index("--from") + 1] if "--from" in sys.argv else "mha_self_fwd"
    a_to = sys.argv[sys.argv.index("--to") + 1] if "--to" in sys.argv else "igemm_pl_wgrad"
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                         r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
    rows.sort()
    stems = [i for i, r in enumerate(rows) if "igemm_stem_kernel" in r[2] or "stem_pl_kernel" in r[2]]
    a, b = stems[-2], stems[-1]
    step = rows[a:b]
    if "--all" in sys.argv:
        i0, i1 = 0, len(step)
    else:
        i0 = next(i for i, r in enumerate(step) if a_from in r[2])
        # (the window ends at the first `--to` kernel BEHIND the last self-attention backward: since round 5 the encoder FFN's
        #  weight gradients are igemm_pl_wgrad launches inside the grounding section)
        last_mha = max((i for i, r in enumerate(step) if "mha_self_bwd" in r[2]), default=i0)
        i1 = next(i for i, r in enumerate(step) if a_to in r[2] and i > max(i0, last_mha if a_to == "igemm_pl_wgrad" else i0))
    t0 = step[i0][0]
    t_end = step[i1 - 1][1] if i1 <= len(step) - 1 else step[-1][1]
    win = [r for r in step if r[0] >= t0 and r[0] < step[min(i1, len(step) - 1)][0]]
    print(f"# window {a_from} .. {a_to}: {(step[min(i1, len(step) - 1)][0] - t0) / 1e3:.1f} us, {len(win)} dispatches "
          f"(step: {len(step)} dispatches, {(rows[b][0] - step[0][0]) / 1e6:.2f} ms)")
    last_end = {}
    per = defaultdict(lambda: [0, 0])
    perq = defaultdict(lambda: [0, 0])
    for s, e, n, q, g, w in win:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        if "--quiet" not in sys.argv:
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f} q{q} {g:>8}/{w:<4} {short(n)}")
        per[short(n)][0] += 1
        per[short(n)][1] += e - s
        perq[q][0] += 1
        perq[q][1] += e - s
    print("# per kernel (count, total us, avg us)")
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"#  {c:5d} {t / 1e3:9.1f} {t / 1e3 / c:7.1f}  {n}")
    for q, (c, t) in perq.items():
        print(f"# queue {q}: {c} dispatches, {t / 1e3:.1f} us of kernels")


if __name__ == "__main__":
    main()
