#!/bin/bash
# Round-5 GPU pass F: hardware-reciprocal GELU + two loader waves + the 2^48 row split: kernel / UNet / VAE parity, ff_tail timing, UNet latency
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5f}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_rowchain_gpu.py tests/test_unet_gpu.py tests/test_vae_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_pytest.log 2>&1; el "pytest exit $? : $(tail -1 $O/${P}_pytest.log)"
grep -h "max-abs\|headroom\|worst\|vae decode 4\|FAILED\|Error" $O/${P}_pytest.log | head -40
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "geglu" > $O/${P}_geglu.log 2>&1; el "geglu kernel tests exit $? : $(tail -1 $O/${P}_geglu.log)"
SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_rctiming.so timeout 300 python tools/ff_tail_timing.py > $O/${P}_timing.txt 2>&1; el "ff_tail_timing exit $?"
grep -v amdgpu $O/${P}_timing.txt | grep "^ABL 2\|^ABL 1"
timeout 300 python tools/bench_ff_tail.py 50 2>&1 | grep -v amdgpu | head -8
for r in 1 2; do
  SDMI_FF_TAIL=0 timeout 300 python tools/unet_latency.py "three launches" 20 2 2>&1 | grep -v amdgpu
  timeout 300 python tools/unet_latency.py "ff_tail, 2 loaders" 20 2 2>&1 | grep -v amdgpu
done
el done
