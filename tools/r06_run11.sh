#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06k
mkdir -p $O
timeout 600 python tools/node_times.py --force-comm > $O/node_times_comm.log 2>&1
timeout 600 python tools/node_times.py > $O/node_times_plain.log 2>&1
timeout 900 python -m pytest tests/test_ops.py tests/test_model_parity.py -m gpu -x -q -k "mha or c3_replayed_bench_step or c3_train_mode_bench_step_against" > $O/tests.log 2>&1
grep -v amdgpu $O/node_times_comm.log; grep -v amdgpu $O/node_times_plain.log; tail -3 $O/tests.log
