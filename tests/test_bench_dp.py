"""bench.py's multi-process path on the GPU box: two ranks (sharing the single GPU, gloo transport — RCCL refuses
duplicate devices) run the full step with the bucketed gradient reducer, the per-rank profiled step and the
final max-over-ranks timing.  Guards against rank-asymmetric collectives (a hang) in the N>1 path."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _release_parent_gpu_memory():
    """These tests start bench.py in child processes on the SAME GPU.  By the time they run, the pytest process' caching
    allocator holds whatever the full-size model tests left cached (well over 100 GB after the C3 / C5 cases); two C3 ranks
    with the reference-sized message beside that once ran out of memory (round 4, exit code 1 of a child).  Hand it back."""
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    yield


def _child_errors(stderr: str) -> str:
    """the ranks' own tracebacks (torchrun's summary at the end of stderr hides them)"""
    lines = stderr.splitlines()
    keep = [i for i, l in enumerate(lines) if "Error" in l or "error" in l or "Traceback" in l]
    out = []
    for i in keep[:12]:
        out.extend(lines[max(0, i - 1):i + 4])
    return "\n".join(out)[-6000:] or stderr[-3000:]


@pytest.mark.gpu
def test_bench_two_ranks_gloo_c3():
    """the benchmark configuration ITSELF (C3: T=64, 448 x 448) under two ranks — both share the box's one
    GPU over gloo, so the step time says nothing; what is checked is the N > 1 contract of the line: n_gpus = world size,
    value = videos of ALL ranks per max-over-ranks step, weak scaling, the 327 MB exchange, an exposed-communication figure."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, STCAT_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-exact", "--no-optim", "--no-profile"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, _child_errors(r.stderr)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    # N > 1 exchanges the reference-sized message by default (327 MB of hot-path gradients + the 498 MB stand-in for the
    # RoBERTa gradients the reference's DDP also reduces) and times the hot path's own 327 MB beside it
    assert d["config"]["workload"].startswith("C3") and d["config"]["allreduce_bytes"] == 327341100 + 4 * 124_645_632
    assert d["hot_path_only_exchange"]["allreduce_bytes"] == 327341100 and d["hot_path_only_exchange"]["value"] > 0
    assert d["roofline"] is None or d["roofline"]["frac"] > 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]
    assert d["exposed_comm_ms_per_step"] is not None and d["dtype"].startswith("f32-class")


@pytest.mark.gpu
def test_bench_default_line_over_the_rccl_path_single_rank():
    """ONE run of the default command line (all side measurements on) at C1 with STCAT_FORCE_COMM=1, which drives the
    complete RCCL path (process group on the device, barriers, bucketed async all-reduce + wait + mean, the loss's
    box-count all-reduce inside every step) with one rank — what a 1-GPU box can check of the N>1 configuration:
      * the bench line is the LAST stdout line although RCCL prints a banner through C stdio;
      * the default reports the fp32-class arithmetic (three bf16 planes, six cross terms),
        carries the 16-bit throughput mode beside it, never switches launch modes behind the caller's back, and pays the
        per-step weight-plane refresh inside the timed steps;
      * round 4: the loss plan is rebuilt inside every timed step; the roofline carries the serialised per-launch figure."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, STCAT_FORCE_COMM="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    env.pop("STCAT_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--config", "C1",
           "--no-cpu-baseline", "--no-optim", "--roberta-dummy"]   # reference-sized message: 824 MB
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    d = json.loads(last)                       # raises if anything (e.g. the RCCL banner) follows the line
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["allreduce_bytes"] > 8.2e8
    assert d["exposed_comm_ms_per_step"] is not None and d["exposed_comm_ms_per_step"] >= 0
    assert d["dtype"].startswith("f32-class") and "launch_modes" not in d
    assert d["config"]["launch"].startswith("launch plans")      # round 3: composite nodes replayed by one C call each
    assert not d["plan_stats"].get("refused"), d["plan_stats"]   # round 6: no node falls back to eager under a live group
    assert d["config"]["loss_plan"].startswith("rebuilt inside every timed step")
    assert d["throughput_mode"]["mma"].startswith("3 bf16 cross terms") and d["throughput_mode"]["steps"] == 10
    assert d["exact_f32_mode"]["steps"] == 10 and d["exact_f32_mode"]["warmup"] == 5   # (five with the pipelined prefix)
    # round 4: the experimental 22-bit mode (two fp16 planes) is timed beside the headline, never in its place
    assert d["near_f32_mode"]["mma"].startswith("3 fp16 cross terms") and d["near_f32_mode"]["steps"] == 10
    assert d["kernels"]["stcat_weight_planes_multi"]["launches"] == 1
    assert d["roofline"]["mfma_flops_per_algorithmic_flop"] == 6
    assert d["roofline"]["isolated_avg_launch_ms"] > 0 and 0 < d["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_bench_graph_option_still_runs():
    """`bench.py --graph` (whole-step hipGraph, opt-in because it measured slower on ROCm 7.2) must keep working:
    capture with device-resident dropout counters, zero-arena memset inside the graph, gradient exchange outside."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--graph", "--steps", "2", "--warmup", "1", "--config", "C1",
           "--no-cpu-baseline", "--no-exact", "--no-optim", "--no-profile"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d["value"] > 0 and d["config"]["launch"] == "one hipGraph per step"
