#!/bin/bash
# Round artifacts on the GPU box: full gpu test suite, smoke, bench line, rocprofv3 kernel stats, PMC HBM traffic.
TAG=${1:-r01}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|mismatch|gradient" | cut -c1-1500 | tee gpurun_out/$TAG/tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee gpurun_out/$TAG/smoke.log
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/$TAG/bench_line.json | cut -c1-300
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$TAG -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact --no-profile > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$TAG -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact --no-profile > /dev/null 2>&1
cd $R
cp /tmp/prof_$TAG/*kernel_stats.csv gpurun_out/$TAG/bench_kernel_stats.csv 2>/dev/null
python tools/pmc_traffic.py /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG > gpurun_out/$TAG/hbm_traffic.json
head -c 1500 gpurun_out/$TAG/hbm_traffic.json
