#!/bin/bash
# Round 6, pass F: re-tune the (tile, split-K) table on the round-6 kernels / precision allocation (new shapes: the 3-pass convs of the last ResBlock,
# single-pass stream 1x1 convs at the 16x16 / 8x8 levels, K-concatenated context K / V), same-box A/B against the committed table, goldens with it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6f}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python tools/tune.py --out $O/${P}_tune.txt --dump $O/${P}_tune_dump.txt --workloads unet64,unet96 --rounds 72 --reps 4 > $O/${P}_tune.log 2>&1; el "tune exit $?: $(tail -1 $O/${P}_tune.log)"
diff <(grep -v "^#" stable-diffusion_amd/tune_gfx950.txt | cut -d' ' -f1-10) <(grep -v "^#" $O/${P}_tune.txt | cut -d' ' -f1-10) | grep -c "^>" 
for rep in 1 2 3; do
  timeout 300 python tools/unet_latency.py "committed table" 20 2 2>&1 | grep round
  SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 300 python tools/unet_latency.py "re-tuned table" 20 2 2>&1 | grep round
done > $O/${P}_lat.log 2>&1; el "latency exit $?"; cat $O/${P}_lat.log
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 1500 python -m pytest tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet (re-tuned) exit $? : $(tail -1 $O/${P}_unet.log)"
grep "^.\?\[unet" $O/${P}_unet.log | cut -c1-120
for w in txt2img768; do
  timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/${P}_b768_old.log 2>&1; tail -1 $O/${P}_b768_old.log | cut -c1-120
  SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/${P}_b768_new.log 2>&1; tail -1 $O/${P}_b768_new.log | cut -c1-120
done
el done
