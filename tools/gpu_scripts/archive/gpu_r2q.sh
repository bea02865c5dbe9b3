#!/bin/bash
# side stream for the ResBlock skip convolution: parity + same-box A/B (SDMI_SIDE_STREAM=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-q}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py tests/test_sampler_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet/pipeline/sampler tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -E "^FAILED|^ERROR" $O/${P}_unet.log | head; grep -E "\[unet .*max-abs" $O/${P}_unet.log | sed 's/^[.F]*//' | cut -c1-110 | head -14
for i in 1 2 3; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_side$i.log 2>&1; el "bench (side stream) exit $?"; tail -1 $O/${P}_bench_side$i.log | cut -c1-130
SDMI_SIDE_STREAM=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_single$i.log 2>&1; el "bench (single stream) exit $?"; tail -1 $O/${P}_bench_single$i.log | cut -c1-130
done
el done
