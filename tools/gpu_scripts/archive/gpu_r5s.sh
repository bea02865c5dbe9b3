#!/bin/bash
# Round-5 GPU pass S: gn_conv3 with the XCD-cooperative weight prefetch: parity, cold-operand bench, UNet latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5s}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for pf in 0 1; do
  SDMI_GN_CONV_PF=$pf timeout 300 python -m pytest tests/test_gnconv_gpu.py -x -q -m gpu -p no:cacheprovider > $O/${P}_gnconv_$pf.log 2>&1; rc=$?; el "gnconv tests PF=$pf exit $rc : $(tail -1 $O/${P}_gnconv_$pf.log)"
  if [ $rc -ne 0 ]; then tail -30 $O/${P}_gnconv_$pf.log | cut -c1-200; exit 1; fi
done
for pf in 0 1; do echo PF=$pf; SDMI_GN_CONV_PF=$pf timeout 300 python tools/bench_gn_conv3_cold.py 30 2>&1 | grep -v amdgpu; done | tee $O/${P}_cold.txt
for r in 1 2; do
  SDMI_GN_CONV=0 timeout 300 python tools/unet_latency.py "gn_conv off" 20 2 2>&1 | grep -v amdgpu
  SDMI_GN_CONV_PF=0 timeout 300 python tools/unet_latency.py "gn_conv PF=0" 20 2 2>&1 | grep -v amdgpu
  SDMI_GN_CONV_PF=1 timeout 300 python tools/unet_latency.py "gn_conv PF=1" 20 2 2>&1 | grep -v amdgpu
done
el done
