#!/bin/bash
# Round-5 GPU pass Y: the split-K reduction that applies the consuming GroupNorm behind a grid barrier: kernel tests, UNet tests, latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5y}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 120 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -s -k "grid_barrier" > $O/${P}_kernels.log 2>&1; rc=$?; el "kernel tests exit $rc : $(tail -1 $O/${P}_kernels.log)"
grep -h "grid barrier" $O/${P}_kernels.log | cut -c1-200 | head
if [ $rc -ne 0 ]; then tail -40 $O/${P}_kernels.log | cut -c1-220; exit 1; fi
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; rc=$?; el "unet tests exit $rc : $(tail -1 $O/${P}_unet.log)"
grep -h "headroom\|grid barrier" $O/${P}_unet.log | cut -c1-200
if [ $rc -ne 0 ]; then tail -40 $O/${P}_unet.log | cut -c1-220; exit 1; fi
for r in 1 2; do
  SDMI_REDUCE_GN_COOP=0 timeout 300 python tools/unet_latency.py "reduce + GroupNorm-apply launches" 20 2 2>&1 | grep -v amdgpu
  timeout 300 python tools/unet_latency.py "GroupNorm inside the reduction" 20 2 2>&1 | grep -v amdgpu
done
timeout 150 python tools/prof_shapes.py 2>&1 | grep "^total\|groupnorm\|splitk_reduce"
SDMI_REDUCE_GN_COOP=0 timeout 150 python tools/prof_shapes.py 2>&1 | grep "^total\|groupnorm\|splitk_reduce"
el done
