#!/bin/bash
# GEMM-family microbench (bf16x3): experiment libraries (timing only, wrong results) vs the product library
for lib in stcat_amd/lib/libstcat_exp*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  STCAT_LIB_OVERRIDE=$PWD/$lib timeout 200 python tools/bench_gemm.py --mma bf16x3 2>&1 | tail -13 | grep -E "TOTAL|l3.conv2|l2.conv2|l3.conv1|l1.conv2" | cut -c1-150
done
if [ "$1" != "exp-only" ]; then
echo "== product build"
timeout 200 python tools/bench_gemm.py --mma bf16x3 2>&1 | tail -13 | cut -c1-150
fi
