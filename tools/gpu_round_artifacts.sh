#!/bin/bash
# Round artifacts on the GPU box: bench line, rocprofv3 kernel stats, PMC HBM traffic + MFMA utilisation (own passes).
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/$TAG/bench_line.json; cut -c1-400 gpurun_out/$TAG/bench_line.json
R=$PWD; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-exact --no-optim"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$TAG -o p -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$TAG -o p -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m_$TAG -o p -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2>&1
cd $R
cp /tmp/prof_$TAG/*kernel_stats.csv gpurun_out/$TAG/bench_kernel_stats.csv 2>/dev/null
python tools/pmc_traffic.py /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG > gpurun_out/$TAG/hbm_traffic.json
python tools/pmc_mfma_util.py /tmp/pmc_m_$TAG > gpurun_out/$TAG/mfma_util.json
head -c 1200 gpurun_out/$TAG/mfma_util.json; echo; head -c 800 gpurun_out/$TAG/hbm_traffic.json
