#!/bin/bash
# PMC passes over one shape of the GEMM microbench.  usage: gpu_probe_pmc.sh <shape-substring> <counter-set ...>
R=$PWD; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
if [ ! -f $R/gpurun_out/counters.txt ]; then rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1; fi
SHAPE="$1"; shift
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python $R/tools/bench_gemm.py --mma bf16x3 --only "$SHAPE" > /tmp/pmc_$i.log 2>&1 || tail -5 /tmp/pmc_$i.log
  echo "== $set"; (cd $R; python tools/pmc_summary.py /tmp/pmc_$i | grep -i "igemm" | cut -c1-400)
done
