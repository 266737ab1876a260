#!/bin/bash
# self-attention at the C3 shape: time + MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs))
timeout 100 python tools/bench_attn.py 207
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_a -o p -- python $R/tools/bench_attn.py 207 > /dev/null 2>&1
cd $R; python tools/pmc_summary.py /tmp/pmc_a
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_a/**/*counter_collection.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
# the first 22 launches of mha_self_fwd_kernel<7> are the inference-mode timing loop (no stash), the rest keep P
per = collections.defaultdict(list)
for r in rows:
    if 'mha_self_fwd_kernel<7>' in r['Kernel_Name']:
        per[(r['Dispatch_Id'])].append((r['Counter_Name'], float(r['Counter_Value'])))
ids = sorted(per, key=int)
def util(i):
    d = dict(per[i]); return d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024)
print('mha_self_fwd<7> MFMA utilisation per launch: first (with stash) %.3f ... inference launches: %s' % (
    util(ids[0]), ' '.join('%.3f' % util(i) for i in ids[1:6])))
PY
