#!/bin/bash
# Round-5 GPU pass L: gn_conv3 (GroupNorm + SiLU + conv3x3 as one launch, a workgroup owns all output columns): parity, per-launch bench, UNet parity + latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5l}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests/test_gnconv_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_gnconv.log 2>&1; rc=$?; el "gnconv tests exit $rc : $(tail -1 $O/${P}_gnconv.log)"
grep -h "vs launches\|max-abs" $O/${P}_gnconv.log | cut -c1-200 | head -30
if [ $rc -ne 0 ]; then tail -40 $O/${P}_gnconv.log | cut -c1-250; exit 1; fi
timeout 300 python tools/bench_gn_conv3.py 50 2>&1 | grep -v amdgpu | tee $O/${P}_bench_gn_conv3.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "max-abs\|headroom" $O/${P}_unet.log | grep -v batch | cut -c1-200 | head -14
for r in 1 2; do
  SDMI_GN_CONV=0 timeout 300 python tools/unet_latency.py "gn_conv off" 20 2 2>&1 | grep -v amdgpu
  timeout 300 python tools/unet_latency.py "gn_conv on" 20 2 2>&1 | grep -v amdgpu
done
el done
