#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06m
mkdir -p $O
timeout 1500 python bench.py 2> $O/bench.err | tail -1 > $O/bench_line.json
python -c "
import json
d=json.loads(open('$O/bench_line.json').read())
r=d['roofline']
print(d['ms_per_step'], d['value'], r['traffic_detail'].get('stale'), r['mfma_util'].get('stale'), r.get('traffic_GB_per_step'), r.get('traffic_over_algorithmic_fp32'), r.get('traffic_over_algorithmic_planes'))
"
