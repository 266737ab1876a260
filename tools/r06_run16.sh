#!/bin/bash
# round 6, GPU call 16: the background prefix in pieces of a few frames (fewer workgroups per launch than CUs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06p
mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
for i in 1 2; do
  run r0_$i STCAT_PREFIX_RANGE=0
  run r8_$i STCAT_PREFIX_RANGE=8
  run r4_$i STCAT_PREFIX_RANGE=4
  run r2_$i STCAT_PREFIX_RANGE=2
  run off_$i STCAT_NO_PREFIX_PIPELINE=1
done
STCAT_PREFIX_RANGE=4 timeout 600 python tools/node_times.py > $O/node_times_r4.log 2>&1
STCAT_PREFIX_RANGE=2 timeout 600 python tools/node_times.py > $O/node_times_r2.log 2>&1
STCAT_PREFIX_RANGE=4 timeout 900 python -m pytest tests/test_plans.py -m gpu -x -q -k "prefix" > $O/tests_r4.log 2>&1
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; grep -v amdgpu $O/node_times_r4.log; grep -v amdgpu $O/node_times_r2.log | head -8; tail -3 $O/tests_r4.log
