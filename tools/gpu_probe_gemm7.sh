#!/bin/bash
timeout 120 python tools/bench_gemm.py --mma bf16x1 2>&1 | tail -14 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9}'
