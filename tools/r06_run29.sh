#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06z
timeout 600 python -m pytest tests/test_library.py -m gpu -x -q > gpurun_out/r06z/tests.log 2>&1; tail -5 gpurun_out/r06z/tests.log
