#!/bin/bash
# Round-4 GPU pass I: re-tune of the (tile, split-K) table on the final kernels (register-order write-through slabs make split launches cheaper):
# unet64 + unet96 shapes, A/B against the committed table on both workloads, parity of every UNet golden with the new table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-i}
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
cp $L/tune_gfx950.txt $O/${P}_tune.txt
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 900 python tools/tune.py --workloads unet64,unet96 --rounds 72 --reps 4 --out $O/${P}_tune.txt --dump $O/${P}_tune_dump.txt > $O/${P}_tune.log 2>&1; el "tune unet64,unet96 exit $? : $(tail -1 $O/${P}_tune.log)"
diff <(grep -v "^#" $L/tune_gfx950.txt | cut -d" " -f1-10 | sort) <(grep -v "^#" $O/${P}_tune.txt | cut -d" " -f1-10 | sort) | grep "^[<>]" | head -60
for r in 1 2 3; do
  timeout 300 python tools/unet_latency.py "committed table" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 300 python tools/unet_latency.py "re-tuned table" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B 64x64"; cat $O/${P}_ab.txt
timeout 600 python bench.py --workload txt2img768 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_b768_old.log 2>&1; el "768 committed: $(tail -1 $O/${P}_b768_old.log | cut -c60-125)"
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 600 python bench.py --workload txt2img768 --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_b768_new.log 2>&1; el "768 re-tuned: $(tail -1 $O/${P}_b768_new.log | cut -c60-125)"
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet goldens (re-tuned) exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "\[unet \|headroom\|^FAILED" $O/${P}_unet.log | sed 's/^\.*//' | cut -c1-150 | head -20
el done
