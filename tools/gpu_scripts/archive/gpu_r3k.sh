#!/bin/bash
# Round-3 GPU pass K: the five-wave 64 x 160 tile (csrc/igemm5.hip, tile 22): kernel tests, re-tune of the bench workload's shapes
# with it among the candidates (residual prefetch off again), parity + A/B + bench + per-shape table with the new table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "five_wave" > $O/k_kern.log 2>&1; el "five-wave kernel tests exit $? : $(tail -1 $O/k_kern.log)"
grep -h "^FAILED\|Error\|^\[igemm5 sd" $O/k_kern.log | cut -c1-200 | head -20
cp $L/tune_gfx950.txt $O/k_tune.txt
SDMI_TUNE_FILE=$O/k_tune.txt timeout 600 python tools/tune.py --workloads unet64 --rounds 96 --reps 4 --out $O/k_tune.txt --dump $O/k_tune_dump.txt > $O/k_tune.log 2>&1; el "tune unet64 exit $? : $(tail -1 $O/k_tune.log)"
echo "entries with tile 22: $(awk '$9==22' $O/k_tune.txt | wc -l)"; awk '$9==22' $O/k_tune.txt | head -30
for r in 1 2; do
  timeout 300 python tools/unet_latency.py "committed table" 20 2 2>/dev/null | grep round >> $O/k_ab.txt
  SDMI_TUNE_FILE=$O/k_tune.txt timeout 300 python tools/unet_latency.py "re-tuned table (tile 22 allowed)" 20 2 2>/dev/null | grep round >> $O/k_ab.txt
done
el "A/B tables"; cat $O/k_ab.txt
SDMI_TUNE_FILE=$O/k_tune.txt timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider > $O/k_unet.log 2>&1; el "unet tests (re-tuned table) exit $? : $(tail -1 $O/k_unet.log)"
grep -h "^\[unet \|headroom\|^FAILED" $O/k_unet.log | cut -c1-170 | head -30
SDMI_TUNE_FILE=$O/k_tune.txt timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/k_bench.log 2>&1; el "bench (re-tuned) exit $?"; tail -1 $O/k_bench.log | cut -c1-330
SDMI_TUNE_FILE=$O/k_tune.txt SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/k_shapes.txt 2>&1; el "prof_shapes exit $?"; grep -v amdgpu $O/k_shapes.txt | head -45 | cut -c1-140
el done
