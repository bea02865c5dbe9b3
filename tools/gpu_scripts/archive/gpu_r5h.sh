#!/bin/bash
# Round-5 GPU pass H: st_mid (out-projection of attn1 + to_q of attn2 as one launch): parity tests, UNet parity, UNet latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5h}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests/test_rowchain_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_rowchain.log 2>&1; rc=$?; el "rowchain tests exit $rc : $(tail -1 $O/${P}_rowchain.log)"
grep -h "vs launches" $O/${P}_rowchain.log | cut -c1-250 | head -30
if [ $rc -ne 0 ]; then tail -40 $O/${P}_rowchain.log | cut -c1-250; fi
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "max-abs\|headroom" $O/${P}_unet.log | grep -v batch | cut -c1-200 | head -14
for r in 1 2; do
  SDMI_ST_MID=0 SDMI_ST_HEAD=0 SDMI_FF_TAIL=0 timeout 300 python tools/unet_latency.py "no chains" 20 2 2>&1 | grep -v amdgpu
  SDMI_ST_MID=0 timeout 300 python tools/unet_latency.py "st_mid off" 20 2 2>&1 | grep -v amdgpu
  timeout 300 python tools/unet_latency.py "st_mid on" 20 2 2>&1 | grep -v amdgpu
done
el done
