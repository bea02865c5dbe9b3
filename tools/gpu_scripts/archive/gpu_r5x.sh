#!/bin/bash
# Round-5 GPU pass X: the pruned product library (experiments compiled out): whole GPU suite; the experiments library: its marked tests;
# UNet latency of both (same box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5x}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/${P}_pytest.log 2>&1; el "product library: pytest -m gpu exit $? : $(tail -1 $O/${P}_pytest.log)"
SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_exp.so timeout 900 python -m pytest tests -x -q -m "gpu and experiments" -p no:cacheprovider > $O/${P}_pytest_exp.log 2>&1; el "experiments library: pytest -m 'gpu and experiments' exit $? : $(tail -1 $O/${P}_pytest_exp.log)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; el "smoke exit $?"; grep smoke: $O/${P}_smoke.log
for r in 1 2; do
  timeout 300 python tools/unet_latency.py "product library" 20 2 2>&1 | grep -v amdgpu
  SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_exp.so timeout 300 python tools/unet_latency.py "experiments library" 20 2 2>&1 | grep -v amdgpu
done
el done
