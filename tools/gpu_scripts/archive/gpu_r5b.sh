#!/bin/bash
# Round-5 GPU pass B: phase stamps and timing-only ablations of ff_tail_kernel (tools/ff_tail_timing.py on the -DSDMI_RC_TIMING build).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5b}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_rctiming.so timeout 300 python tools/ff_tail_timing.py > $O/${P}_timing.txt 2>&1; el "ff_tail_timing exit $?"
grep -v amdgpu $O/${P}_timing.txt
el done
