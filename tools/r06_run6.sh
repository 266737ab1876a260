#!/bin/bash
# round 6, GPU call 6: bit-exact staged prefix; probed CU-masked / low-priority lanes; per-kernel stats of the attention A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06f
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_plans.py -m gpu -x -q -k "prefix" > $O/tests_prefix.log 2>&1
echo "rc=$?" >> $O/tests_prefix.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/bench_$name.json 2> $O/bench_$name.err; }
for i in 1 2; do
  run norm_$i A=1
  run cu224_$i STCAT_PREFIX_CUS=224
  run cu192_$i STCAT_PREFIX_CUS=192
  run cu128_$i STCAT_PREFIX_CUS=128
  run low_$i STCAT_PREFIX_PRIO=1
done
STCAT_PREFIX_CUS=192 timeout 600 python tools/node_times.py > $O/node_times_cu192.log 2>&1
for v in bs6 fp32pipe; do
  if [ $v = fp32pipe ]; then export STCAT_MHA_FP32_PIPE=1; else unset STCAT_MHA_FP32_PIPE; fi
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_$v -o st -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact --no-optim --no-profile > $OLDPWD/$O/prof_$v.log 2>&1)
done
unset STCAT_MHA_FP32_PIPE
for v in bs6 fp32pipe; do f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep -i "mha\|attn" $f) > $O/stats_$v.csv; done
for f in $O/bench_*.json; do python -c "
import sys, json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d.get('prefix_lane'))
except Exception as e: print('$f', 'FAILED', e)
"; done > $O/summary.txt 2>&1
cat $O/summary.txt; tail -5 $O/tests_prefix.log; cat $O/stats_bs6.csv $O/stats_fp32pipe.csv; grep -v amdgpu $O/node_times_cu192.log
# keep the merged output small
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
