#!/bin/bash
# A/B of HIP runtime environment knobs on the step (C3 and C1 = host floor) and on the captured-decoder eval path.
O=gpurun_out/${1:-envp}; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-optim --no-profile --steps 10 --warmup 3"
P='import json,sys; d=json.loads(sys.stdin.readline()); e=d.get("eval_path") or {}; print({k:d.get(k) for k in ("ms_per_step","host_enqueue_ms_per_step")}, "eval", e.get("ms_per_video"), (e.get("captured_decoder") or {}).get("ms_per_video"))'
{
for v in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"; do
  echo "## C3 $v"; env $v timeout 400 $B 2>&1 | tail -1 | python -c "$P"
done
for v in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "## C1 $v"; env $v timeout 300 $B --no-exact --config C1 2>&1 | tail -1 | python -c "$P"
done
} > $O/env.log 2>&1; cat $O/env.log
