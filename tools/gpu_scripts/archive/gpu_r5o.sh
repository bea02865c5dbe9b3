#!/bin/bash
# Round-5 GPU pass O: gn_conv3 with two chunks of look-ahead for the fp32 halo: parity, per-launch bench, UNet latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5o}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for v in "4 0" "4 1"; do
  set -- $v
  SDMI_GN_CONV_NS=$1 SDMI_GN_CONV_PF=$2 timeout 300 python -m pytest tests/test_gnconv_gpu.py -x -q -m gpu -p no:cacheprovider > $O/${P}_gnconv_$1_$2.log 2>&1; rc=$?; el "gnconv tests NS=$1 PF=$2 exit $rc : $(tail -1 $O/${P}_gnconv_$1_$2.log)"
  if [ $rc -ne 0 ]; then tail -30 $O/${P}_gnconv_$1_$2.log | cut -c1-200; exit 1; fi
done
timeout 300 python tools/bench_gn_conv3.py 50 2>&1 | grep -v amdgpu | head -8 | tee $O/${P}_bench_gn_conv3.txt
for r in 1 2; do
  SDMI_GN_CONV=0 timeout 300 python tools/unet_latency.py "gn_conv off" 20 2 2>&1 | grep -v amdgpu
  SDMI_GN_CONV_PF=0 timeout 300 python tools/unet_latency.py "gn_conv PF=0" 20 2 2>&1 | grep -v amdgpu
  SDMI_GN_CONV_PF=1 timeout 300 python tools/unet_latency.py "gn_conv PF=1" 20 2 2>&1 | grep -v amdgpu
done
SDMI_GN_CONV=1 SDMI_GN_CONV_PF=0 timeout 150 python tools/prof_shapes.py 2>&1 | grep "gnconv\|^total\|groupnorm"
el done
