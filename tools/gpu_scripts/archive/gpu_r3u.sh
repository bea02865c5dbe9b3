#!/bin/bash
# Round-3 GPU pass U: GroupNorm of the SpatialTransformer applied inside the proj_in GEMM (gemm_split16_gn_kernel, SDMI_GN_PROJ_FOLD):
# bit-identity + goldens, same-box A/B, per-shape times; and the per-workgroup phase stamps of the round-3 build (timing library).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_unet_gpu.py -q -s -p no:cacheprovider -k "proj_in or golden or headroom" > $O/u_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/u_unet.log)"
grep -o "\[unet headroom[^[]*" $O/u_unet.log | head -2; grep -h "^FAILED\|AssertionError" $O/u_unet.log | head
for r in 1 2; do
  SDMI_GN_PROJ_FOLD=0 timeout 300 python tools/unet_latency.py "GroupNorm-apply launch + proj_in GEMM" 20 2 2>/dev/null | grep round >> $O/u_ab.txt
  timeout 300 python tools/unet_latency.py "GroupNorm inside the proj_in GEMM" 20 2 2>/dev/null | grep round >> $O/u_ab.txt
done
el "A/B"; cat $O/u_ab.txt
for f in 0 1; do SDMI_GN_PROJ_FOLD=$f SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py 2>/dev/null | grep "split16\|groupnorm\|^total" | cut -c1-120 | sed "s/^/FOLD=$f  /"; done
if [ -f $L/libsdmi_timing.so ]; then
  SDMI_GN_PROJ_FOLD=0 SDMI_LIB_PATH=$L/libsdmi_timing.so timeout 600 python tools/igemm_timing.py $O/u_timing.txt > $O/u_timing.log 2>&1; el "phase stamps exit $?"; grep -c . $O/u_timing.txt
fi
el done
