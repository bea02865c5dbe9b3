#!/bin/bash
# Round-5 GPU pass E: ff_tail with 1 / 2 / 4 loader waves (parity, phase stamps, microbenchmark, UNet A/B) + localising the > 2 GB decode bug
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5e}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for ld in 1 2 4; do
  SDMI_FF_TAIL_LD=$ld timeout 300 python -m pytest tests/test_rowchain_gpu.py -x -q -m gpu -p no:cacheprovider > $O/${P}_rowchain_ld$ld.log 2>&1; el "rowchain tests (loaders $ld) exit $? : $(tail -1 $O/${P}_rowchain_ld$ld.log)"
done
SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_rctiming.so timeout 300 python tools/ff_tail_timing.py > $O/${P}_timing.txt 2>&1; el "ff_tail_timing exit $?"
grep -v amdgpu $O/${P}_timing.txt
for ld in 1 2 4; do
  echo "== loaders $ld"; SDMI_FF_TAIL_LD=$ld timeout 300 python tools/bench_ff_tail.py 50 2>&1 | grep -v amdgpu | head -8
done
for r in 1 2; do
  for ld in 1 2 4; do SDMI_FF_TAIL_LD=$ld timeout 300 python tools/unet_latency.py "ff_tail loaders $ld" 20 2 2>&1 | grep -v amdgpu; done
done
el "2 GB: default"; timeout 300 python tools/dbg_2gb.py 4 2>&1 | grep -v amdgpu | tee $O/${P}_2gb_default.txt
el "2 GB: SDMI_EPI_VEC=0"; SDMI_EPI_VEC=0 timeout 300 python tools/dbg_2gb.py 4 2>&1 | grep -v amdgpu | tee $O/${P}_2gb_epivec0.txt
el "2 GB: no tuning table"; SDMI_TUNE_DISABLE=1 timeout 300 python tools/dbg_2gb.py 4 2>&1 | grep -v amdgpu | tee $O/${P}_2gb_notune.txt
el done
