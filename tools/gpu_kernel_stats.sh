#!/bin/bash
# rocprofv3 per-kernel summary of a short bench run -> gpurun_out/kernel_stats.csv (+ top list on stdout)
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact --no-optim > /dev/null 2>&1
cd $R; mkdir -p gpurun_out; cp /tmp/ks/*kernel_stats.csv gpurun_out/kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/kernel_stats.csv')))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/step', tot / 5e6)
for r in rows[:45]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls'])//5:5d}/step {float(r['TotalDurationNs'])/5e6:7.3f} ms  avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
