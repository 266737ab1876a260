#!/bin/bash
timeout 120 python tools/bench_gemm.py --mma bf16x3 2>&1 | tail -14
timeout 120 python tools/bench_gemm.py --mma bf16x3 --tile 128x64 2>&1 | tail -14
timeout 120 python tools/bench_gemm.py --mma bf16x3 --tile 64x64 2>&1 | tail -14
