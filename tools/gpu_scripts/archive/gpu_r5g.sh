#!/bin/bash
# Round-5 GPU pass G: st_head chain kernel: parity tests, microbenchmark + phase stamps, UNet parity with it on, UNet latency A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5g}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests/test_rowchain_gpu.py -x -q -m gpu -p no:cacheprovider -s -k st_head > $O/${P}_rowchain.log 2>&1; rc=$?; el "st_head tests exit $rc : $(tail -1 $O/${P}_rowchain.log)"
grep -h "st_head" $O/${P}_rowchain.log | cut -c1-250 | head -30
if [ $rc -ne 0 ]; then tail -40 $O/${P}_rowchain.log | cut -c1-250; fi
timeout 300 python tools/bench_st_head.py 50 2>&1 | grep -v amdgpu | tee $O/${P}_bench.txt
SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_rctiming.so timeout 300 python tools/bench_st_head.py 20 2>&1 | grep "wave-0" | tee -a $O/${P}_bench.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "max-abs\|headroom" $O/${P}_unet.log | grep -v batch | cut -c1-200
for r in 1 2; do
  SDMI_ST_HEAD=0 timeout 300 python tools/unet_latency.py "st_head off" 20 2 2>&1 | grep -v amdgpu
  timeout 300 python tools/unet_latency.py "st_head on" 20 2 2>&1 | grep -v amdgpu
done
el done
