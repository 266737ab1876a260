#!/bin/bash
timeout 120 python tools/bench_gemm.py --mma bf16x3 2>&1 | tail -14
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -k "bf16" 2>&1 | tail -2
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/bench_gemm.py --mma bf16x3 --only "l3.conv2" > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc2 -o p -- python $R/tools/bench_gemm.py --mma bf16x3 --only "l3.conv2" > /dev/null 2>&1
cd $R
python tools/pmc_summary.py /tmp/pmc1 /tmp/pmc2
