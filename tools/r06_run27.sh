#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06x
mkdir -p $O
timeout 900 python -m pytest tests/test_plans.py tests/test_loader.py tests/test_model_parity.py -m gpu -x -q -k "prefix or prefetcher or train_mode_bench_step_against or c3_replayed_bench_step or two_forwards" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile 2> $O/bench.err | tail -1 > $O/bench.json
tail -3 $O/tests.log; python -c "
import json; d=json.loads(open('$O/bench.json').read()); print(d['ms_per_step'], d['plan_stats'], d['config']['prefix_pipeline'][-90:])"
