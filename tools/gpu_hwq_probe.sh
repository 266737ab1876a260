#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; streams that share one serialise.  10-step bench
# lines + device timelines (per-queue summary of tools/timeline.py), plain and with a live RCCL group at one rank.
O=gpurun_out/${1:-hwq}; mkdir -p $O; R=$PWD
B="python $R/bench.py --no-cpu-baseline --no-exact --no-optim --no-profile"
P='import json,sys; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ("ms_per_step","host_enqueue_ms_per_step","exposed_comm_ms_per_step","streams")})'
{
for v in "X=1" "STCAT_FORCE_COMM=1" "STCAT_FORCE_COMM=1 STCAT_RCCL_NORMAL_PRIORITY=1"; do
  echo "## $v"; env $v timeout 300 $B --steps 10 --warmup 3 2>&1 | tail -1 | python -c "$P"
done
} > $O/variants.log 2>&1; cat $O/variants.log
cd /tmp; export TMPDIR=/tmp
i=0
for v in "X=1" "STCAT_FORCE_COMM=1"; do
  i=$((i+1))
  env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlq_$i -o t -- $B --steps 3 --warmup 2 > /dev/null 2>&1
  { echo "## $v"; python $R/tools/timeline.py /tmp/tlq_$i; } > $R/$O/timeline_$i.log 2>&1
  head -6 $R/$O/timeline_$i.log; grep -A6 "hardware queues" $R/$O/timeline_$i.log
done
