#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06t
mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 900 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run auto A=1
run eager STCAT_PREFIX_EAGER=1
run r0 STCAT_PREFIX_RANGE=0
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['other_modes'].items()})
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -x -q -k "train_mode_bench_step_against_fixture" > $O/tests.log 2>&1; tail -3 $O/tests.log
