#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06r
mkdir -p $O
timeout 900 python -m pytest tests/test_plans.py tests/test_loader.py -m gpu -x -q -k "prefix or prefetcher" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
STCAT_PREFIX_RANGE=4 timeout 900 python -m pytest tests/test_plans.py -m gpu -x -q -k "prefix" > $O/tests_r4.log 2>&1; echo "rc=$?" >> $O/tests_r4.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
for i in 1 2; do
  run r0_$i STCAT_PREFIX_RANGE=0
  run r4_$i STCAT_PREFIX_RANGE=4
  run r0comm_$i STCAT_PREFIX_RANGE=0 STCAT_FORCE_COMM=1
  run r4comm_$i STCAT_PREFIX_RANGE=4 STCAT_FORCE_COMM=1
  run off_$i STCAT_NO_PREFIX_PIPELINE=1
  run offcomm_$i STCAT_NO_PREFIX_PIPELINE=1 STCAT_FORCE_COMM=1
done
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['plan_stats'])
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; tail -3 $O/tests.log; tail -3 $O/tests_r4.log
