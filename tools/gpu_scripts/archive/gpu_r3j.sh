#!/bin/bash
# Round-3 GPU pass J: residual quads of the small-tile epilogues fetched in the kernel prologue (SDMI_EPI_PREFETCH), non-temporal
# output stores (libsdmi_nt.so, -DSDMI_NT_STORES), re-tune of the bench workload's shapes on this build, parity + A/B + bench with it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider > $O/j_kern.log 2>&1; el "kernel tests exit $? : $(tail -1 $O/j_kern.log)"
timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider > $O/j_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/j_unet.log)"
grep -h "^\[unet \|headroom\|^FAILED" $O/j_unet.log | cut -c1-170 | head -30
for r in 1 2; do
  SDMI_EPI_PREFETCH=0 timeout 300 python tools/unet_latency.py "residual in the epilogue (PREFETCH=0)" 20 2 2>/dev/null | grep round >> $O/j_ab.txt
  timeout 300 python tools/unet_latency.py "residual prefetched" 20 2 2>/dev/null | grep round >> $O/j_ab.txt
  SDMI_LIB_PATH=$L/libsdmi_nt.so timeout 300 python tools/unet_latency.py "non-temporal output stores" 20 2 2>/dev/null | grep round >> $O/j_ab.txt
done
el "A/B"; cat $O/j_ab.txt
cp $L/tune_gfx950.txt $O/j_tune.txt
SDMI_TUNE_FILE=$O/j_tune.txt timeout 600 python tools/tune.py --workloads unet64 --rounds 72 --reps 4 --out $O/j_tune.txt --dump $O/j_tune_dump.txt > $O/j_tune.log 2>&1; el "tune unet64 exit $? : $(tail -1 $O/j_tune.log)"
for r in 1 2; do
  timeout 300 python tools/unet_latency.py "committed table" 20 2 2>/dev/null | grep round >> $O/j_ab2.txt
  SDMI_TUNE_FILE=$O/j_tune.txt timeout 300 python tools/unet_latency.py "re-tuned table" 20 2 2>/dev/null | grep round >> $O/j_ab2.txt
done
el "A/B tables"; cat $O/j_ab2.txt
SDMI_TUNE_FILE=$O/j_tune.txt timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider > $O/j_unet2.log 2>&1; el "unet tests (re-tuned table) exit $? : $(tail -1 $O/j_unet2.log)"
grep -h "^\[unet \|headroom\|^FAILED" $O/j_unet2.log | cut -c1-170 | head -30
SDMI_TUNE_FILE=$O/j_tune.txt timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/j_bench.log 2>&1; el "bench (re-tuned) exit $?"; tail -1 $O/j_bench.log | cut -c1-330
el done
