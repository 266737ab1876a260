#!/bin/bash
# round 6, GPU call 2: prefix queued at the query decoder's entry — tests, same-box A/B against "behind the backbone" and off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06b
mkdir -p $O
timeout 900 python -m pytest tests/test_plans.py -m gpu -x -q -k "prefix" > $O/prefix_tests.log 2>&1
echo "prefix tests rc=$?" >> $O/prefix_tests.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
for i in 1 2; do
  timeout 600 $B > $O/bench_dec_$i.json 2> $O/bench_dec_$i.err
  STCAT_PREFIX_AT=backbone timeout 600 $B > $O/bench_bb_$i.json 2> $O/bench_bb_$i.err
  timeout 600 $B --no-prefix-pipeline > $O/bench_off_$i.json 2> $O/bench_off_$i.err
done
timeout 600 python tools/node_times.py > $O/node_times_dec.log 2>&1
for f in $O/bench_*.json; do python -c "
import sys, json
d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'])
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; tail -3 $O/prefix_tests.log; cat $O/node_times_dec.log
