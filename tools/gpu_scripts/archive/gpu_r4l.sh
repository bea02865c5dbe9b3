#!/bin/bash
# Round-4 GPU pass L: the merged tuning table (pass I's choices for the 512 / 768 workloads' shapes, pass J's for the rest) against pass J's
# all-workload table on one box, then the evidence sequence of gpu_r4c.sh on the merged (committed) table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-l}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for r in 1 2 3; do
  SDMI_TUNE_FILE=$PWD/tools/gpu_scripts/.j_tune_r4.txt timeout 300 python tools/unet_latency.py "pass-J table (tuned on a slow box)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  timeout 300 python tools/unet_latency.py "merged table (I for unet64/96 shapes)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B"; cat $O/${P}_ab.txt
for w in txt2img768; do
  SDMI_TUNE_FILE=$PWD/tools/gpu_scripts/.j_tune_r4.txt timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_b_$w_j.log 2>&1; el "bench $w J: $(tail -1 $O/${P}_b_$w_j.log | cut -c60-125)"
  timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_b_$w_m.log 2>&1; el "bench $w merged: $(tail -1 $O/${P}_b_$w_m.log | cut -c60-125)"
done
bash tools/gpu_scripts/gpu_r4c.sh $P
