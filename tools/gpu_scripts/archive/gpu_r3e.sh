#!/bin/bash
# Round-3 GPU pass E (re-entry after the container was re-created; the logs of passes B-D were lost): localise the intermittent
# mismatch of the GroupNorm-folding conv with four builds / knobs on ONE box (default, every counted wait = vmcnt(0), compiler-
# visible fp32 loads, a full barrier behind every conversion, no raw copy), then the state of the default build: kernel tests of
# the fold + split-fp16 families, UNet goldens, same-box A/B of the knobs, one bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
st() { grep -h "differ\|TOTAL\|it [0-9]" $1 | cut -c1-230 | head -${2:-40}; }
timeout 300 python tools/gn_fold_stress.py --iters 16 --cases w64,w32 > $O/e_stress.log 2>&1; el "stress default exit $?"; st $O/e_stress.log 60
SDMI_GN_SAFE=1 timeout 300 python tools/gn_fold_stress.py --iters 16 --cases w64 > $O/e_stress_safe.log 2>&1; el "stress SAFE exit $?"; st $O/e_stress_safe.log
SDMI_LIB_PATH=$L/libsdmi_gnvis.so timeout 300 python tools/gn_fold_stress.py --iters 16 --cases w64 > $O/e_stress_vis.log 2>&1; el "stress visible-loads build exit $?"; st $O/e_stress_vis.log
SDMI_LIB_PATH=$L/libsdmi_gnxbar.so timeout 300 python tools/gn_fold_stress.py --iters 16 --cases w64 > $O/e_stress_xbar.log 2>&1; el "stress extra-barrier build exit $?"; st $O/e_stress_xbar.log
timeout 300 python tools/gn_fold_stress.py --iters 16 --cases w64 --raw 0 > $O/e_stress_noraw.log 2>&1; el "stress no-raw exit $?"; st $O/e_stress_noraw.log
timeout 300 python tools/gn_fold_stress.py --iters 16 --cases w64 --flush 0 > $O/e_stress_noflush.log 2>&1; el "stress no-flush exit $?"; st $O/e_stress_noflush.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "gn_fold or split16 or split_fp16" > $O/e_kern.log 2>&1; el "fold + split16 kernel tests exit $? : $(tail -1 $O/e_kern.log)"
grep -h "^FAILED" $O/e_kern.log | cut -c1-150
run() { env "$@" timeout 300 python tools/unet_repeat.py --case sdv1_64x64 --reps 5 2>&1 | grep "rep " | cut -c1-120; }
el "unet default"; run SDMI_FUSE_GN_CONV=1
el "unet every fold site as two launches"; run SDMI_GN_FORCE_TWO=1
el "unet every fold site in the fold kernel"; run SDMI_GN_FORCE_TWO=-1
for r in 1 2; do
  SDMI_FUSE_GN_CONV=0 SDMI_SPLIT16_KERNEL=0 timeout 300 python tools/unet_latency.py "fold0 s16-0 (round-2 path)" 20 2 2>/dev/null | grep round >> $O/e_ab.txt
  SDMI_FUSE_GN_CONV=0 SDMI_SPLIT16_KERNEL=1 timeout 300 python tools/unet_latency.py "fold0 s16-1" 20 2 2>/dev/null | grep round >> $O/e_ab.txt
  SDMI_GN_FORCE_TWO=1 timeout 300 python tools/unet_latency.py "fold sites two-launch s16-1" 20 2 2>/dev/null | grep round >> $O/e_ab.txt
  timeout 300 python tools/unet_latency.py "default (fold heuristic) s16-1" 20 2 2>/dev/null | grep round >> $O/e_ab.txt
  SDMI_GN_FORCE_TWO=-1 timeout 300 python tools/unet_latency.py "fold kernel everywhere s16-1" 20 2 2>/dev/null | grep round >> $O/e_ab.txt
done
el "A/B"; cat $O/e_ab.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider > $O/e_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/e_unet.log)"
grep -h "^\[unet \|headroom\|^FAILED" $O/e_unet.log | cut -c1-160 | head -30
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/e_bench.log 2>&1; el "bench exit $?"; tail -1 $O/e_bench.log | cut -c1-500
el done
