#!/bin/bash
# fused split-K reduction: kernel tests, same-box A/B against the separate reduce kernel (SDMI_SPLITK_FUSED=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-o}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x > $O/${P}_kernels.log 2>&1; el "kernel tests exit $? : $(tail -1 $O/${P}_kernels.log)"
grep -E "^FAILED|^ERROR" $O/${P}_kernels.log | head -30
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_clip_gpu.py tests/test_vae_gpu.py tests/test_pipeline_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet/clip/vae/pipeline tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -E "^FAILED|^ERROR" $O/${P}_unet.log | head; grep -E "\[unet .*max-abs" $O/${P}_unet.log | sed 's/^[.F]*//' | head -14
timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_fused.txt 2>&1; el "prof_shapes (fused) exit $?"; grep -E "^total|^splitk" $O/${P}_shapes_fused.txt
SDMI_SPLITK_FUSED=0 timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_unfused.txt 2>&1; el "prof_shapes (unfused) exit $?"; grep -E "^total|^splitk" $O/${P}_shapes_unfused.txt
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_fused$i.log 2>&1; el "bench (fused) exit $?"; tail -1 $O/${P}_bench_fused$i.log | cut -c1-200
SDMI_SPLITK_FUSED=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_unfused$i.log 2>&1; el "bench (unfused) exit $?"; tail -1 $O/${P}_bench_unfused$i.log | cut -c1-200
done
el done
