#!/usr/bin/env python3
"""GPU box: run bench.py's train-mode step at C3 (its own step object, pipelined prefix, launch plans: eager, eager,
record, REPLAY) and write the dropout stream of the replayed step — seed, device base, (offset, decisions) of every site —
to gpurun_out/c3_train_trace.json.  In the build container `tests/golden/make_golden.py train C3 <that file>` feeds these
masks to the CPU oracle (fp32 + fp64) and writes tests/golden/model_C3_train.npz; the GPU test
test_gpu_c3_train_mode_bench_step_against_fixture asserts that its live stream equals the stored one before it compares."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from tests import test_model_parity as P  # noqa: E402
from tests.backends import use_hip  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"{name.lower()}_train_trace.json")
    dev = use_hip()
    tr = {}
    keep, losses, grads = P._run_bench_step(dev, name, P.BENCH_MMA, train=True, pipeline=True, trace=tr)
    tr["case"] = name
    tr["losses"] = losses
    tr["pred_sted_absmax"] = float(keep["pred_sted"].abs().max())
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(tr, f)
    print(f"{name}: {len(tr['sites'])} dropout sites, seed {tr['seed']}, base {tr['base']}, total loss {losses['total']:.6f} -> {out}")


if __name__ == "__main__":
    main()
