#!/bin/bash
# round 6, GPU call 3: which lane for the pipelined prefix — least-priority stream, CU-masked streams, default priority
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06c
mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err
}
for i in 1 2; do
  run low_$i STCAT_PREFIX_PRIO=1
  run norm_$i STCAT_PREFIX_PRIO=0
  run cu192_$i STCAT_PREFIX_CUS=192
  run cu128_$i STCAT_PREFIX_CUS=128
  run cu64_$i STCAT_PREFIX_CUS=64
  run off_$i STCAT_NO_PREFIX_PIPELINE=1
done
run lowbb_1 STCAT_PREFIX_PRIO=1 STCAT_PREFIX_AT=backbone
run cu128bb_1 STCAT_PREFIX_CUS=128 STCAT_PREFIX_AT=backbone
STCAT_PREFIX_PRIO=1 timeout 600 python tools/node_times.py > $O/node_times_low.log 2>&1
STCAT_PREFIX_CUS=128 timeout 600 python tools/node_times.py > $O/node_times_cu128.log 2>&1
STCAT_PREFIX_CUS=128 STCAT_PREFIX_AT=backbone timeout 600 python tools/node_times.py > $O/node_times_cu128bb.log 2>&1
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; cat $O/node_times_low.log $O/node_times_cu128.log $O/node_times_cu128bb.log | grep -v amdgpu
