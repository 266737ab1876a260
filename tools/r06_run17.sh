#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06q
mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
for i in 1 2; do
  run r0_$i STCAT_PREFIX_RANGE=0
  run r4_$i STCAT_PREFIX_RANGE=4
  run r3_$i STCAT_PREFIX_RANGE=3
  run r4bb_$i STCAT_PREFIX_RANGE=4 STCAT_PREFIX_AT=backbone
  run r4exit_$i STCAT_PREFIX_RANGE=4 STCAT_PREFIX_TRIGGER=exit
  run r4comm_$i STCAT_PREFIX_RANGE=4 STCAT_FORCE_COMM=1
  run r0comm_$i STCAT_PREFIX_RANGE=0 STCAT_FORCE_COMM=1
done
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt
