#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06l
mkdir -p $O
timeout 1500 python -m pytest tests/test_dp_model.py tests/test_bench_dp.py tests/test_dist.py -m gpu -x -q > $O/tests_dp.log 2>&1
echo "rc=$?" >> $O/tests_dp.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
for i in 1 2; do
  run plain_$i A=1
  run comm_$i STCAT_FORCE_COMM=1
  run comm_off_$i STCAT_FORCE_COMM=1 STCAT_NO_PREFIX_PIPELINE=1
  run comm_dummy_$i STCAT_FORCE_COMM=1 STCAT_X=1
done
timeout 600 python tools/node_times.py --force-comm > $O/node_times_comm.log 2>&1
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['plan_stats'], d['exposed_comm_ms_per_step'])
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; tail -4 $O/tests_dp.log; grep -v "amdgpu\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket\|destroy_process" $O/node_times_comm.log
