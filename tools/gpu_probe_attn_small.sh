#!/bin/bash
# GPU durations of the tiny (one video, T queries) self-attention launches of the decoders / temporal encoder
R=$PWD; cd /tmp; export TMPDIR=/tmp
for S in 64 65; do
  rm -rf /tmp/pa_$S
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_$S -o p -- python $R/tools/bench_attn.py $S 1 > /dev/null 2>&1
  python - <<PY
import csv, glob
f = glob.glob('/tmp/pa_$S/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'mha_self' in r['Name']: print('S=$S', r['Name'][:48], r['Calls'], 'avg %.1f us' % (float(r['AverageNs'])/1e3))
PY
done
