#!/bin/bash
timeout 120 python tools/bench_gemm.py --mma bf16x6 2>&1 | tail -14
timeout 120 python tools/bench_gemm.py --mma bf16x6 --tile 128x64 2>&1 | tail -1
timeout 120 python tools/bench_gemm.py --mma bf16x6 --tile 64x64 2>&1 | tail -1
