#!/bin/bash
timeout 120 python tools/bench_gemm.py --mma bf16x3 --variant 0 2>&1 | tail -1
timeout 120 python tools/bench_gemm.py --mma bf16x3 --variant 1 2>&1 | tail -14
timeout 120 python tools/bench_gemm.py --mma bf16x3 --variant 1 --tile 128x64 2>&1 | tail -14
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -k "bf16x3" 2>&1 | tail -2
