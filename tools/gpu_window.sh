#!/bin/bash
# trace of a few bench steps -> dispatch listing of the grounding window (tools/window_dump.py) + the whole-step summary
TAG=${1:-w}
O=gpurun_out/$TAG
mkdir -p $O
R=$PWD
B="python $R/bench.py --no-cpu-baseline --no-exact --no-optim"
timeout 300 $B --steps 10 --warmup 3 --no-profile 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print({k:d.get(k) for k in ("ms_per_step","host_enqueue_ms_per_step")})' > $O/step.log 2>&1
timeout 300 $B --steps 10 --warmup 3 --no-profile --config C1 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("C1", {k:d.get(k) for k in ("ms_per_step","host_enqueue_ms_per_step")})' >> $O/step.log 2>&1
cat $O/step.log
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o t -- $B --steps 3 --warmup 1 --no-profile > /dev/null 2>&1
cd $R
python tools/timeline.py /tmp/tl_$TAG > $O/timeline.log 2>&1
python tools/window_dump.py /tmp/tl_$TAG > $O/window.log 2>&1
head -8 $O/timeline.log; grep "^#" $O/window.log | head -60
