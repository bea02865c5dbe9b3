#!/bin/bash
# Round-3 GPU pass R: LayerNorm fold extended to 1280 channels / >= 512 token rows (48 -> 3 LayerNorm launches per call): kernel + UNet
# tests, same-box A/B against SDMI_LN_FOLD_MIN_ROWS=2048 (the previous default), re-tune of the bench workload, parity + A/B + bench with it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "layernorm_folded or attention_ctx" > $O/r_kern.log 2>&1; el "ln-fold / attn_ctx kernel tests exit $? : $(tail -1 $O/r_kern.log)"
timeout 600 python -m pytest tests/test_unet_gpu.py -q -s -p no:cacheprovider > $O/r_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/r_unet.log)"
grep -o "\[unet headroom[^[]*" $O/r_unet.log | head -2; grep -h "^FAILED" $O/r_unet.log | head
for r in 1 2; do
  SDMI_LN_FOLD_MIN_ROWS=2048 timeout 300 python tools/unet_latency.py "LN fold from 2048 rows (before)" 20 2 2>/dev/null | grep round >> $O/r_ab.txt
  timeout 300 python tools/unet_latency.py "LN fold from 512 rows, C <= 1280" 20 2 2>/dev/null | grep round >> $O/r_ab.txt
done
el "A/B (committed table)"; cat $O/r_ab.txt
cp $L/tune_gfx950.txt $O/r_tune.txt
SDMI_TUNE_FILE=$O/r_tune.txt timeout 600 python tools/tune.py --workloads unet64 --rounds 96 --reps 4 --out $O/r_tune.txt --dump $O/r_tune_dump.txt > $O/r_tune.log 2>&1; el "tune unet64 exit $? : $(tail -1 $O/r_tune.log)"
for r in 1 2; do
  SDMI_LN_FOLD_MIN_ROWS=2048 timeout 300 python tools/unet_latency.py "fold from 2048, committed table" 20 2 2>/dev/null | grep round >> $O/r_ab2.txt
  SDMI_TUNE_FILE=$O/r_tune.txt timeout 300 python tools/unet_latency.py "fold from 512, re-tuned table" 20 2 2>/dev/null | grep round >> $O/r_ab2.txt
done
el "A/B (re-tuned)"; cat $O/r_ab2.txt
SDMI_TUNE_FILE=$O/r_tune.txt timeout 600 python -m pytest tests/test_unet_gpu.py -q -s -p no:cacheprovider -k "golden or headroom or TUNE" > $O/r_unet2.log 2>&1; el "unet goldens (re-tuned) exit $? : $(tail -1 $O/r_unet2.log)"
grep -o "\[unet [^]]*\][^[]*" $O/r_unet2.log | cut -c1-120 | tail -14
SDMI_TUNE_FILE=$O/r_tune.txt SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/r_shapes.txt 2>&1; el "prof_shapes exit $?"; grep -v amdgpu $O/r_shapes.txt | head -8 | cut -c1-120
el done
