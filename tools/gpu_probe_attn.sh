#!/bin/bash
timeout 900 python -m pytest tests/test_model_parity.py -m gpu -q -k "c2 or c5 or eval_path" 2>&1 | grep -E "passed|failed|Error" | cut -c1-600
timeout 100 python tools/bench_attn.py 207
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_a -o p -- python $R/tools/bench_attn.py 207 > /dev/null 2>&1
cd $R; python tools/pmc_summary.py /tmp/pmc_a
