#!/bin/bash
# Round-5 GPU pass T: the XCD-cooperative weight-prefetch wave in the row-strip chain kernels (ff_tail / st_tail / st_head / st_mid) and in
# gn_conv3 (+ residual prefetch): parity, UNet parity, UNet latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5t}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for pf in 1 0; do
  SDMI_CHAIN_PF=$pf SDMI_GN_CONV_PF=$pf timeout 300 python -m pytest tests/test_rowchain_gpu.py tests/test_gnconv_gpu.py -x -q -m gpu -p no:cacheprovider > $O/${P}_chain_$pf.log 2>&1; rc=$?; el "chain + gnconv tests PF=$pf exit $rc : $(tail -1 $O/${P}_chain_$pf.log)"
  if [ $rc -ne 0 ]; then tail -30 $O/${P}_chain_$pf.log | cut -c1-200; exit 1; fi
done
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "headroom" $O/${P}_unet.log | cut -c1-200
for r in 1 2; do
  SDMI_GN_CONV=0 SDMI_CHAIN_PF=0 timeout 300 python tools/unet_latency.py "chain PF=0, gn_conv off" 20 2 2>&1 | grep -v amdgpu
  SDMI_GN_CONV=0 SDMI_CHAIN_PF=1 timeout 300 python tools/unet_latency.py "chain PF=1, gn_conv off" 20 2 2>&1 | grep -v amdgpu
  SDMI_GN_CONV=1 SDMI_CHAIN_PF=1 SDMI_GN_CONV_PF=1 timeout 300 python tools/unet_latency.py "chain PF=1, gn_conv PF=1" 20 2 2>&1 | grep -v amdgpu
done
el done
