#!/bin/bash
# Round artifacts on the GPU box: full GPU test-suite, smoke, the bench line, rocprofv3 kernel stats, PMC HBM traffic +
# MFMA utilisation (own passes), device timeline, host profile / phase times, step-like GEMM table, comm-mode lines.
# Everything lands under gpurun_out/$TAG; the summaries worth judging are copied into profiles/ afterwards.
TAG=${1:-r06}
O=gpurun_out/$TAG
mkdir -p $O
R=$PWD
if [ "$2" != "notests" ]; then
  # the whole GPU selection, evidence first (tests/conftest.py orders it), durations + the gradient reports kept
  timeout 1500 python -m pytest tests -q -m gpu --durations=15 -rP -p no:cacheprovider > $O/gpu_tests_full.log 2>&1
  grep -E "gradient report|^[0-9.]+s (call|setup)|passed|failed|^FAILED|^ERROR" $O/gpu_tests_full.log | cut -c1-420 > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $O/smoke.log; tail -1 $O/smoke.log
fi
timeout 1200 python bench.py 2>&1 | tail -1 > $O/bench_line.json; cut -c1-300 $O/bench_line.json
B="python $R/bench.py --no-cpu-baseline --no-exact --no-optim"
P='import json,sys; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ("ms_per_step","host_enqueue_ms_per_step","host_cpu_ms_per_step","exposed_comm_ms_per_step","n_gpus")})'
{
  echo "# C3, 10 steps: plain | STCAT_FORCE_COMM=1 (RCCL path, 1 rank) | + --roberta-dummy (824 MB message) | --serial (one stream) | --hoist-loss-plan | C1 (same launches, ~no GPU work: the host floor)"
  timeout 300 $B --steps 10 --warmup 3 --no-profile 2>&1 | tail -1 | python -c "$P"
  STCAT_FORCE_COMM=1 timeout 300 $B --steps 10 --warmup 3 --no-profile 2>&1 | tail -1 | python -c "$P"
  STCAT_FORCE_COMM=1 timeout 300 $B --steps 10 --warmup 3 --no-profile --roberta-dummy 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --steps 10 --warmup 3 --no-profile --serial 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --steps 10 --warmup 3 --no-profile --hoist-loss-plan 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --steps 10 --warmup 3 --no-profile --no-auto-graph --config C1 2>&1 | tail -1 | python -c "$P"
  echo "# round 6: --no-prefix-pipeline (every step computes its clip's frozen prefix itself) | STCAT_MHA_FP32_PIPE=1 (self-attention on the fp32-pipe kernels of round 5) | default again"
  timeout 300 $B --steps 10 --warmup 3 --no-profile --no-prefix-pipeline 2>&1 | tail -1 | python -c "$P"
  STCAT_MHA_FP32_PIPE=1 timeout 300 $B --steps 10 --warmup 3 --no-profile 2>&1 | tail -1 | python -c "$P"
  timeout 300 $B --steps 10 --warmup 3 --no-profile 2>&1 | tail -1 | python -c "$P"
} > $O/bench_variants.log 2>&1; cat $O/bench_variants.log
# host profile of the bench mode with launch plans ON: the untraced node timeline (events between the autograd nodes of the
# replayed step) + the per-phase table of tools/phase_times.py (default mode = the bench mode) + the Python-side profile
{ echo "## tools/node_times.py (C3, bf16x6p, plans on, untraced: HIP events between the nodes of the replayed step)"
  timeout 300 python tools/node_times.py 2>&1 | grep -v amdgpu.ids | tail -40
  echo "## tools/node_times.py --config C1 (same launches, ~no GPU work)"
  timeout 300 python tools/node_times.py --config C1 2>&1 | grep -v amdgpu.ids | tail -40
  echo "## tools/phase_times.py (bench mode, plans on)"
  timeout 300 python tools/phase_times.py 2>&1 | grep -v amdgpu.ids | tail -12
  echo "## tools/host_profile.py"
  timeout 300 python tools/host_profile.py 2>&1 | grep -v amdgpu.ids | head -40
} > $O/host_profile.log 2>&1; grep -A12 "phase_times" $O/host_profile.log | head -14
timeout 300 python tools/bench_gemm.py --mma bf16x6p --step-like 2>&1 | grep -v amdgpu.ids > $O/plane_gemm_steplike.log; tail -3 $O/plane_gemm_steplike.log
cd /tmp; export TMPDIR=/tmp
# (round 6: --no-profile as well, so that the trace holds NOTHING but headline steps — 5 plan warm-ups (eager, eager with the
#  staged prefix, recorded; the staged prefix's own two plans) + 1 warm-up + 3 timed = 9 identical steps; VERDICT r05 weak #9: the round-5 CSVs also held the
#  two instrumented profiling steps of bench.py)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- $B --steps 3 --warmup 1 --no-profile > /dev/null 2>&1
# the same command on ONE stream: per-kernel durations there are ISOLATED (nothing co-runs) — the figures `roofline` is priced on
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$TAG -o bench -- $B --serial --steps 3 --warmup 1 --no-profile > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$TAG -o p -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$TAG -o p -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m_$TAG -o p -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o t -- $B --steps 3 --warmup 1 --no-profile > /dev/null 2>&1
cd $R
cp /tmp/prof_$TAG/*/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null || cp /tmp/prof_$TAG/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
cp /tmp/profs_$TAG/*/*kernel_stats.csv $O/bench_kernel_stats_serial.csv 2>/dev/null || cp /tmp/profs_$TAG/*kernel_stats.csv $O/bench_kernel_stats_serial.csv 2>/dev/null
python tools/pmc_traffic.py /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG > $O/hbm_traffic.json
python tools/pmc_mfma_util.py /tmp/pmc_m_$TAG > $O/mfma_util.json
python tools/timeline.py /tmp/tl_$TAG > $O/timeline.log
head -c 600 $O/mfma_util.json; echo; head -c 500 $O/hbm_traffic.json; echo; head -5 $O/timeline.log; head -8 $O/bench_kernel_stats.csv | cut -c1-160
# per-tensor gradient error of the bench arithmetic against the oracle's fp64 run at the benchmark size (full tensors; ~3 min of host time)
timeout 1200 python tools/grad_error_report.py --config C3 --modes bf16x6p --out $O/grad_error_C3_bf16x6p.json > $O/grad_error.log 2>&1; grep -E "^==|backbone|ground_|input_proj" $O/grad_error.log | cut -c1-200 | head -30
