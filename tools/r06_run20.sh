#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06s
mkdir -p $O
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile --mma f16x3p"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
run f16_auto A=1
run f16_r0 STCAT_PREFIX_RANGE=0
run f16_off STCAT_NO_PREFIX_PIPELINE=1
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile --mma bf16x3p"
run x3p_auto A=1
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['plan_stats'])
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; tail -5 $O/bench_f16_auto.err
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -x -q -k "train_mode_bench_step_against_fixture" > $O/tests.log 2>&1; tail -3 $O/tests.log
