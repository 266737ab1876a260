#!/bin/bash
# full GPU suite + bench in a given mode:  bash tools/gpu_run_suite.sh [mma] [tag]
MMA=${1:-bf16x3}; TAG=${2:-x}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|mismatch|gradient" | cut -c1-2500 | tee gpurun_out/tests_$TAG.log
timeout 120 python tools/bench_gemm.py --mma $MMA 2>&1 | tail -1
timeout 600 python bench.py --steps 5 --warmup 2 --mma $MMA --cpu-sample-frames 2 2>&1 | tail -1 | tee gpurun_out/bench_$TAG.log | cut -c1-400
