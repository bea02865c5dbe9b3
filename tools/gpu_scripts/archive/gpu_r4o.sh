#!/bin/bash
# Round-4 GPU pass O: a second tune of the headline workloads' shapes (unet64, unet96) on whatever box this is, A/B against the committed (merged)
# table on the 512 and 768 workloads -- is the committed choice stable across boxes of the pool?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-o}
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
cp $L/tune_gfx950.txt $O/${P}_tune.txt
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 900 python tools/tune.py --workloads unet64,unet96 --rounds 72 --reps 6 --out $O/${P}_tune.txt --dump $O/${P}_tune_dump.txt > $O/${P}_tune.log 2>&1; el "tune exit $? : $(tail -1 $O/${P}_tune.log)"
diff <(grep -v "^#" $L/tune_gfx950.txt | cut -d" " -f1-10 | sort) <(grep -v "^#" $O/${P}_tune.txt | cut -d" " -f1-10 | sort) | grep -c "^>" | xargs echo "rows changed:"
for r in 1 2 3; do
  timeout 300 python tools/unet_latency.py "committed (merged) table" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 300 python tools/unet_latency.py "re-tuned on this box" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B 64x64"; cat $O/${P}_ab.txt
timeout 600 python bench.py --workload txt2img768 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_b768_old.log 2>&1; el "768 committed: $(tail -1 $O/${P}_b768_old.log | cut -c60-125)"
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 600 python bench.py --workload txt2img768 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_b768_new.log 2>&1; el "768 re-tuned: $(tail -1 $O/${P}_b768_new.log | cut -c60-125)"
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet goldens (re-tuned) exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "headroom\|^FAILED" $O/${P}_unet.log | sed 's/^\.*//' | cut -c1-150 | head
el done
