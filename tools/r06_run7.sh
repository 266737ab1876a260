#!/bin/bash
# round 6, GPU call 7: the six-product bf16-pipe self-attention in isolation (C3 spatial-layer shape): time, per-kernel stats, MFMA-busy
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06g
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
{
for m in bf16x6p bf16x3 f32; do
  echo "## MMA=$m"; MMA=$m timeout 100 python tools/bench_attn.py 207 2>&1 | grep -v amdgpu
done
echo "## MMA=bf16x6p STCAT_MHA_FP32_PIPE=1 (the round-5 path of the default mode)"
MMA=bf16x6p STCAT_MHA_FP32_PIPE=1 timeout 100 python tools/bench_attn.py 207 2>&1 | grep -v amdgpu
} > $O/bench_attn.log 2>&1
cd /tmp
for v in bs6 fp32pipe; do
  if [ $v = fp32pipe ]; then export STCAT_MHA_FP32_PIPE=1; else unset STCAT_MHA_FP32_PIPE; fi
  MMA=bf16x6p timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka_$v -o p -- python $R/tools/bench_attn.py 207 > /dev/null 2>&1
  MMA=bf16x6p timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pa_$v -o p -- python $R/tools/bench_attn.py 207 > /dev/null 2>&1
  MMA=bf16x6p timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$v -o p -- python $R/tools/bench_attn.py 207 > /dev/null 2>&1
  MMA=bf16x6p timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$v -o p -- python $R/tools/bench_attn.py 207 > /dev/null 2>&1
done
unset STCAT_MHA_FP32_PIPE
cd $R
for v in bs6 fp32pipe; do
  f=$(find /tmp/ka_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$v.csv
  python tools/pmc_mfma_util.py /tmp/pa_$v > $O/mfma_util_$v.json 2>&1
  python tools/pmc_traffic.py /tmp/pf_$v /tmp/pw_$v > $O/hbm_traffic_$v.json 2>&1
done
cat $O/bench_attn.log; for v in bs6 fp32pipe; do grep -i "mha" $O/kernel_stats_$v.csv | cut -c1-200; head -c 1500 $O/mfma_util_$v.json; echo; head -c 1200 $O/hbm_traffic_$v.json; echo; done
