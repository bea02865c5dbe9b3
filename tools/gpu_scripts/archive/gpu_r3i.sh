#!/bin/bash
# Round-3 GPU pass I: LayerNorm fold, second version (partials requested at the top of the consumer, fp32 fold, DPP row sums in the
# producer): kernel tests, UNet tests, same-box A/B against SDMI_LN_FOLD=0, per-shape table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "layernorm_folded" > $O/i_kern.log 2>&1; el "ln-fold kernel tests exit $? : $(tail -1 $O/i_kern.log)"
grep -h "^FAILED\|Error" $O/i_kern.log | cut -c1-170 | head
timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider > $O/i_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/i_unet.log)"
grep -h "^\[unet \|headroom\|^FAILED" $O/i_unet.log | cut -c1-170 | head -30
for r in 1 2; do
  SDMI_LN_FOLD=0 timeout 300 python tools/unet_latency.py "LN launches (SDMI_LN_FOLD=0)" 20 2 2>/dev/null | grep round >> $O/i_ab.txt
  timeout 300 python tools/unet_latency.py "LN folded v2 (M >= 2048)" 20 2 2>/dev/null | grep round >> $O/i_ab.txt
done
el "A/B"; cat $O/i_ab.txt
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/i_shapes.txt 2>&1; el "prof_shapes (fold) exit $?"; grep -v amdgpu $O/i_shapes.txt | head -3
SDMI_LN_FOLD=0 SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/i_shapes0.txt 2>&1; el "prof_shapes (no fold) exit $?"; grep -v amdgpu $O/i_shapes0.txt | head -3
for k in "M8192_N320_K320_k1_m0" "M8192_N960_K320" "M8192_N2560_K320" "M8192_N320_K320_s1" "M2048_N640_K640_k1_m0" "M2048_N1920" "M2048_N5120" "M2048_N640_K640_s1" "layernorm"; do
  echo "fold:    $(grep -h "$k" $O/i_shapes.txt | head -2 | cut -c1-120 | tr '\n' '|')"; echo "no fold: $(grep -h "$k" $O/i_shapes0.txt | head -2 | cut -c1-120 | tr '\n' '|')"
done
el done
