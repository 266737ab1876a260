#!/bin/bash
# plane-GEMM timing experiments + device timeline of the step
mkdir -p gpurun_out/r02
O=gpurun_out/r02/pl_exp_$1.log
{
echo "== baseline"; timeout 200 python tools/bench_gemm.py --mma bf16x3p 2>&1 | grep -v amdgpu | cut -c1-150
echo "== no wgrad atomics (flag 1) + no epilogue memory traffic (flag 2)"; timeout 200 python tools/bench_gemm.py --mma bf16x3p --pl-flags 3 2>&1 | grep -v amdgpu | cut -c1-150
echo "== tile 3 (128x128)"; timeout 200 python tools/bench_gemm.py --mma bf16x3p --pl-tile 3 2>&1 | grep -v amdgpu | cut -c1-150
echo "== tile 1 (256x128)"; timeout 200 python tools/bench_gemm.py --mma bf16x3p --pl-tile 1 2>&1 | grep -v amdgpu | cut -c1-150
echo "== tile 2 (128x256)"; timeout 200 python tools/bench_gemm.py --mma bf16x3p --pl-tile 2 2>&1 | grep -v amdgpu | cut -c1-150
} > $O 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-profile --no-optim > /tmp/tl.log 2>&1
cd $R; python tools/timeline.py /tmp/tl > gpurun_out/r02/timeline_$1.log 2>&1
cat $O; cat gpurun_out/r02/timeline_$1.log
