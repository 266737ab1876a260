#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06y
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=15 -rP -p no:cacheprovider > $O/gpu_tests_full.log 2>&1
grep -E "gradient report|^[0-9.]+s (call|setup)|passed|failed|^FAILED|^ERROR" $O/gpu_tests_full.log | cut -c1-420 > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/smoke.log; tail -1 $O/smoke.log
