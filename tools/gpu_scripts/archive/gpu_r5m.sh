#!/bin/bash
# Round-5 GPU pass M: gn_conv3 with a weight-prefetch wave / a deeper ring: parity, UNet latency A/B of the four variants
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5m}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for v in "4 0" "4 1" "6 0" "6 1"; do
  set -- $v
  SDMI_GN_CONV_NS=$1 SDMI_GN_CONV_PF=$2 timeout 300 python -m pytest tests/test_gnconv_gpu.py -x -q -m gpu -p no:cacheprovider > $O/${P}_gnconv_$1_$2.log 2>&1; el "gnconv tests NS=$1 PF=$2 exit $? : $(tail -1 $O/${P}_gnconv_$1_$2.log)"
done
for r in 1 2; do
  SDMI_GN_CONV=0 timeout 300 python tools/unet_latency.py "gn_conv off" 20 2 2>&1 | grep -v amdgpu
  for v in "4 0" "4 1" "6 0" "6 1"; do
    set -- $v
    SDMI_GN_CONV_NS=$1 SDMI_GN_CONV_PF=$2 timeout 300 python tools/unet_latency.py "gn_conv NS=$1 PF=$2" 20 2 2>&1 | grep -v amdgpu
  done
done
el done
