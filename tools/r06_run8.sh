#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06h
mkdir -p $O
timeout 600 python tools/tail_probe.py 2>&1 | grep -v amdgpu > $O/tail_probe.log
cat $O/tail_probe.log
