#!/bin/bash
# round 6, GPU call 4: decoder chains on high-priority streams under the pipelined prefix; the train-mode fixtures
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06d
mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
for i in 1 2; do
  run norm_$i A=1
  run hp_$i STCAT_DECODER_HP=1
  run off_$i STCAT_NO_PREFIX_PIPELINE=1
done
STCAT_DECODER_HP=1 timeout 600 python tools/node_times.py > $O/node_times_hp.log 2>&1
timeout 1500 python -m pytest tests/test_model_parity.py tests/test_plans.py -m gpu -x -q -k "train_mode_bench_step_against_fixture or prefix" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; grep -v amdgpu $O/node_times_hp.log; tail -30 $O/tests.log
