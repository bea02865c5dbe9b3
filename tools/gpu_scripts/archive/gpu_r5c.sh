#!/bin/bash
# Round-5 GPU pass C: the loader-wave version of ff_tail_kernel: parity tests, phase stamps / ablations (timing build), microbenchmark
# against the three launches, UNet latency A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5c}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests/test_rowchain_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_rowchain.log 2>&1; rc=$?; el "rowchain tests exit $rc : $(tail -1 $O/${P}_rowchain.log)"
grep -h "vs launches\|Error\|error" $O/${P}_rowchain.log | head -20
if [ $rc -ne 0 ]; then tail -30 $O/${P}_rowchain.log; fi
SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_rctiming.so timeout 300 python tools/ff_tail_timing.py > $O/${P}_timing.txt 2>&1; el "ff_tail_timing exit $?"
grep -v amdgpu $O/${P}_timing.txt
timeout 300 python tools/bench_ff_tail.py 50 > $O/${P}_ff_tail_bench.txt 2>&1; el "bench_ff_tail exit $?"; grep -v amdgpu $O/${P}_ff_tail_bench.txt | head -16
for r in 1 2; do
  SDMI_FF_TAIL=0 timeout 300 python tools/unet_latency.py "three launches" 20 2 2>&1 | grep -v amdgpu
  SDMI_FF_TAIL=1 timeout 300 python tools/unet_latency.py "ff_tail chain" 20 2 2>&1 | grep -v amdgpu
done
el done
