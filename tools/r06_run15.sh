#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06n
mkdir -p $O
timeout 900 python -m pytest tests/test_library.py tests/test_ops.py -m gpu -x -q > $O/tests_ops.log 2>&1; echo "rc=$?" >> $O/tests_ops.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile 2> $O/bench.err | tail -1 > $O/bench.json
tail -3 $O/tests_ops.log; tail -1 $O/smoke.log; python -c "
import json; d=json.loads(open('$O/bench.json').read()); print(d['ms_per_step'], d['plan_stats'])"
