#!/bin/bash
# Round 6, pass C: precision allocation by measured sensitivity (the last ResBlock's 3x3 convs 3-pass split-fp16, the stream 1x1 convs of the deepest
# level + middle block single-pass): every UNet golden, then same-box timing of the old and new allocations through the experiments library.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6c}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1500 python -m pytest tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet exit $? : $(tail -1 $O/${P}_unet.log)"
grep "^.\?\[unet" $O/${P}_unet.log | cut -c1-250
X=$PWD/stable-diffusion_amd/libsdmi_exp.so
for rep in 1 2; do
  SDMI_LIB_PATH=$X SDMI_PRECISE_LAST_RES=0 SDMI_PRECISE_1X1_MAX_DS=99 SDMI_PRECISE_KV=0 timeout 300 python tools/unet_latency.py "r5 allocation" 20 3 2>&1 | grep round
  SDMI_LIB_PATH=$X SDMI_PRECISE_LAST_RES=0 SDMI_PRECISE_1X1_MAX_DS=8 timeout 300 python tools/unet_latency.py "1x1 single-pass at ds>=8" 20 3 2>&1 | grep round
  SDMI_LIB_PATH=$X SDMI_PRECISE_LAST_RES=0 SDMI_PRECISE_1X1_MAX_DS=4 timeout 300 python tools/unet_latency.py "1x1 single-pass at ds>=4" 20 3 2>&1 | grep round
  SDMI_LIB_PATH=$X SDMI_PRECISE_LAST_RES=1 SDMI_PRECISE_1X1_MAX_DS=8 timeout 300 python tools/unet_latency.py "last res 3-pass, ds>=8" 20 3 2>&1 | grep round
  timeout 300 python tools/unet_latency.py "product library" 20 3 2>&1 | grep round
done > $O/${P}_lat.log 2>&1; el "latency exit $?"; cat $O/${P}_lat.log
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes.log 2>&1; grep "^total\|K17280\|K25920\|K51840\|M128_N1280_K2560\|M128_N1280_K1280" $O/${P}_shapes.log | cut -c1-150
el done
