#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06j
mkdir -p $O
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default bench rc=$?" > $O/summary.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
for i in 1 2; do
  run entry_$i A=1
  run exit_$i STCAT_PREFIX_TRIGGER=exit
  run comm_pipe_$i STCAT_FORCE_COMM=1
  run comm_off_$i STCAT_FORCE_COMM=1 STCAT_NO_PREFIX_PIPELINE=1
  run comm_lane1_$i STCAT_FORCE_COMM=1 STCAT_PREFIX_STREAM=0
done
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['streams'].get('picked'), d['streams'].get('concurrent_with_main'))
except Exception as e: print('$f', 'FAILED', e)
"; done >> $O/summary.txt 2>&1
cat $O/summary.txt; tail -3 $O/bench_default.err
