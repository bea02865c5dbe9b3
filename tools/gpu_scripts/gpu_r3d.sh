#!/bin/bash
# Round-3 GPU pass D: localise the intermittent mismatch of the GroupNorm-folding conv (pass B / C: one W = 64 case per run):
# kernel-level repetition against the two-launch path with cache flushes (where is the difference?), the same with every counted
# wait replaced by vmcnt(0) (is it a wait?), and the SD-v1 64x64 golden repeated under the bisecting knobs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python tools/gn_fold_stress.py --iters 12 --cases w64,w32,w16 > $O/d_stress.log 2>&1; el "stress exit $?"; tail -40 $O/d_stress.log | cut -c1-220
SDMI_GN_SAFE=1 timeout 600 python tools/gn_fold_stress.py --iters 12 --cases w64 > $O/d_stress_safe.log 2>&1; el "stress SAFE exit $?"; grep -h "differ\|TOTAL" $O/d_stress_safe.log | cut -c1-200
timeout 300 python tools/gn_fold_stress.py --iters 12 --cases w64 --raw 0 > $O/d_stress_noraw.log 2>&1; el "stress no-raw exit $?"; grep -h "differ\|TOTAL" $O/d_stress_noraw.log | cut -c1-200
run() { env "$@" timeout 300 python tools/unet_repeat.py --case sdv1_64x64 --reps 6 2>&1 | grep "rep " | cut -c1-120; }
el "unet default (fold kernel + two-launch defaults)"; run SDMI_FUSE_GN_CONV=1
el "unet all two-launch";  run SDMI_GN_FORCE_TWO=1
el "unet all fold kernel"; run SDMI_GN_FORCE_TWO=-1
el "unet fold kernel, in_layers only"; run SDMI_GN_FORCE_TWO=-1 SDMI_FUSE_GN_WHICH=1
el "unet fold kernel, out_layers only"; run SDMI_GN_FORCE_TWO=-1 SDMI_FUSE_GN_WHICH=2
el "unet fold kernel, W=64 only"; run SDMI_GN_FORCE_TWO=-1 SDMI_FUSE_GN_W=64
el "unet fold kernel, W=32 only"; run SDMI_GN_FORCE_TWO=-1 SDMI_FUSE_GN_W=32
el "unet fold kernel, SAFE waits"; run SDMI_GN_FORCE_TWO=-1 SDMI_GN_SAFE=1
el "unet unfused"; run SDMI_FUSE_GN_CONV=0
el done
