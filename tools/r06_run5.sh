#!/bin/bash
# round 6, GPU call 5: six-product bf16-pipe self-attention — kernel tests, model parity at C3, same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06e
mkdir -p $O
timeout 900 python -m pytest tests/test_ops.py -m gpu -x -q -k "mha" > $O/tests_mha.log 2>&1
echo "rc=$?" >> $O/tests_mha.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
for i in 1 2; do
  run bs6_$i A=1
  run fp32pipe_$i STCAT_MHA_FP32_PIPE=1
done
timeout 2400 python -m pytest tests/test_model_parity.py tests/test_plans.py -m gpu -x -q -k "c3 or C3 or prefix or train_mode" > $O/tests_model.log 2>&1
echo "rc=$?" >> $O/tests_model.log
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; tail -5 $O/tests_mha.log; tail -15 $O/tests_model.log
