#!/bin/bash
# round 6, GPU call 1: prefix pipeline — parity tests, same-box A/B, node timeline, train-mode dropout traces
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06a
mkdir -p $O
timeout 900 python -m pytest tests/test_plans.py -m gpu -x -q -k "prefix" > $O/prefix_tests.log 2>&1
echo "prefix tests rc=$?" >> $O/prefix_tests.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
for i in 1 2; do
  timeout 600 $B > $O/bench_pipe_$i.json 2> $O/bench_pipe_$i.err
  timeout 600 $B --no-prefix-pipeline > $O/bench_nopipe_$i.json 2> $O/bench_nopipe_$i.err
done
timeout 600 python tools/node_times.py > $O/node_times_pipe.log 2>&1
timeout 600 python tools/node_times.py --no-prefix-pipeline > $O/node_times_nopipe.log 2>&1
timeout 600 python tools/node_times.py --config C1 > $O/node_times_pipe_c1.log 2>&1
timeout 900 python tools/c3_train_trace.py C3 > $O/trace_c3.log 2>&1
timeout 600 python tools/c3_train_trace.py C1 > $O/trace_c1.log 2>&1
grep -h '"value"' $O/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['config'].get('prefix_pipeline','')[:40])
" > $O/summary.txt 2>&1
cat $O/summary.txt; tail -3 $O/prefix_tests.log; tail -2 $O/trace_c3.log
