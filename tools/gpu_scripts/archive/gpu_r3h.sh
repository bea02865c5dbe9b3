#!/bin/bash
# Round-3 GPU pass H: LayerNorm folded into the consuming GEMM (IGemmParams::lnp_out / lnf_*): kernel tests, UNet goldens, same-box
# A/B against SDMI_LN_FOLD=0, re-tune of the bench workload's shapes (the producers' keys changed: split pinned to 1), parity and
# A/B with the new table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "layernorm_folded" > $O/h_kern.log 2>&1; el "ln-fold kernel tests exit $? : $(tail -1 $O/h_kern.log)"
grep -h "^\[ln-fold\|^FAILED\|Error" $O/h_kern.log | cut -c1-170 | head -50
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider > $O/h_kern_all.log 2>&1; el "all kernel tests exit $? : $(tail -1 $O/h_kern_all.log)"
timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider > $O/h_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/h_unet.log)"
grep -h "^\[unet \|headroom\|^FAILED" $O/h_unet.log | cut -c1-170 | head -30
for r in 1 2; do
  SDMI_LN_FOLD=0 timeout 300 python tools/unet_latency.py "LN launches (SDMI_LN_FOLD=0)" 20 2 2>/dev/null | grep round >> $O/h_ab.txt
  timeout 300 python tools/unet_latency.py "LN folded (M >= 2048)" 20 2 2>/dev/null | grep round >> $O/h_ab.txt
done
el "A/B"; cat $O/h_ab.txt
cp $L/tune_gfx950.txt $O/h_tune.txt
SDMI_TUNE_FILE=$O/h_tune.txt timeout 600 python tools/tune.py --workloads unet64 --rounds 72 --reps 3 --out $O/h_tune.txt --dump $O/h_tune_dump.txt > $O/h_tune.log 2>&1; el "tune unet64 exit $? : $(tail -1 $O/h_tune.log)"
diff <(sort $L/tune_gfx950.txt) <(sort $O/h_tune.txt) | grep -c "^>" 
for r in 1 2; do
  timeout 300 python tools/unet_latency.py "LN folded, committed table" 20 2 2>/dev/null | grep round >> $O/h_ab2.txt
  SDMI_TUNE_FILE=$O/h_tune.txt timeout 300 python tools/unet_latency.py "LN folded, re-tuned table" 20 2 2>/dev/null | grep round >> $O/h_ab2.txt
done
el "A/B tables"; cat $O/h_ab2.txt
SDMI_TUNE_FILE=$O/h_tune.txt timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider > $O/h_unet2.log 2>&1; el "unet tests (re-tuned table) exit $? : $(tail -1 $O/h_unet2.log)"
grep -h "^\[unet \|headroom\|^FAILED" $O/h_unet2.log | cut -c1-170 | head -30
SDMI_TUNE_FILE=$O/h_tune.txt SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/h_shapes.txt 2>&1; el "prof_shapes exit $?"; grep -v amdgpu $O/h_shapes.txt | head -30 | cut -c1-150
el done
