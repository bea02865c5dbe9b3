#!/bin/bash
# Round-5 GPU pass AA: the cross-attention inside the st_mid chain launch (st_head_kernel CTX): parity tests, UNet parity, UNet latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5aa}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests/test_rowchain_gpu.py -x -q -m gpu -p no:cacheprovider -s -k "ctx or cross" > $O/${P}_rowchain.log 2>&1; rc=$?; el "rowchain ctx tests exit $rc : $(tail -1 $O/${P}_rowchain.log)"
grep -h "st_mid_ctx" $O/${P}_rowchain.log | cut -c1-220 | head -14
if [ $rc -ne 0 ]; then tail -40 $O/${P}_rowchain.log | cut -c1-250; exit 1; fi
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; rc=$?; el "unet tests exit $rc : $(tail -1 $O/${P}_unet.log)"
grep -h "headroom" $O/${P}_unet.log | cut -c1-200
if [ $rc -ne 0 ]; then tail -30 $O/${P}_unet.log | cut -c1-250; exit 1; fi
for r in 1 2; do
  SDMI_ST_MID_CTX=0 timeout 300 python tools/unet_latency.py "st_mid + attention launches" 20 2 2>&1 | grep -v amdgpu
  timeout 300 python tools/unet_latency.py "cross-attention inside st_mid" 20 2 2>&1 | grep -v amdgpu
done
el done
