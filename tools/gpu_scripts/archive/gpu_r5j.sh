#!/bin/bash
# Round-5 GPU pass J: st_tail without the carried epilogue indices: parity, per-launch bench, UNet latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5j}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests/test_rowchain_gpu.py -x -q -m gpu -p no:cacheprovider -s -k "tail" > $O/${P}_rowchain.log 2>&1; rc=$?; el "rowchain tests exit $rc : $(tail -1 $O/${P}_rowchain.log)"
if [ $rc -ne 0 ]; then tail -40 $O/${P}_rowchain.log | cut -c1-250; exit 1; fi
timeout 300 python tools/bench_st_tail.py 50 2>&1 | grep -v amdgpu | tee $O/${P}_bench_st_tail.txt
for r in 1 2; do
  SDMI_ST_TAIL=0 timeout 300 python tools/unet_latency.py "st_tail off" 20 2 2>&1 | grep -v amdgpu
  timeout 300 python tools/unet_latency.py "st_tail on" 20 2 2>&1 | grep -v amdgpu
done
el done
