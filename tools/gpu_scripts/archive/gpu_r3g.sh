#!/bin/bash
# Round-3 GPU pass G: (1) the GroupNorm-folding conv with per-class counted waits (LDS-DMA / register loads counted apart):
# UNet golden repeated, fold at every site; (2) the CFG pair as two independent B = 1 chains on two streams (tools/two_stream.py).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
run() { env SDMI_FUSE_GN_CONV=1 SDMI_GN_FORCE_TWO=-1 "$@" timeout 300 python tools/unet_repeat.py --case sdv1_64x64 --reps 10 2>&1 | grep "rep " | cut -c1-120; }
el "unet fold everywhere, per-class waits"; run X=1 | tee $O/g_fold.txt
el "same, second process"; run X=1 | tee -a $O/g_fold.txt
timeout 300 python tools/two_stream.py 20 > $O/g_two_stream.txt 2>&1; el "two-stream exit $?"; grep -v amdgpu.ids $O/g_two_stream.txt
SDMI_FUSE_GN_CONV=1 SDMI_GN_FORCE_TWO=-1 timeout 300 python tools/unet_latency.py "fold everywhere, per-class waits" 20 2 2>/dev/null | grep round
timeout 300 python tools/unet_latency.py "default" 20 2 2>/dev/null | grep round
el done
