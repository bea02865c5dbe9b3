#!/bin/bash
# attention: LDS-DMA ring kernel vs the register-staged one (SDMI_ATTN_V1=1), same box: tests, per-shape table, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-l}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "attention or attn" > $O/${P}_attn.log 2>&1; el "attention kernel tests exit $? : $(tail -1 $O/${P}_attn.log)"
grep -E "^FAILED|^ERROR" $O/${P}_attn.log | head -30
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_clip_gpu.py tests/test_vae_gpu.py tests/test_pipeline_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet/clip/vae/pipeline tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -E "^FAILED|^ERROR" $O/${P}_unet.log | head; grep -E "\[unet .*max-abs" $O/${P}_unet.log | sed 's/^[.F]*//' | head -24
timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_dma.txt 2>&1; el "prof_shapes (dma) exit $?"; grep -E "^total|^attn" $O/${P}_shapes_dma.txt
SDMI_ATTN_V1=1 timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_v1.txt 2>&1; el "prof_shapes (v1) exit $?"; grep -E "^total|^attn" $O/${P}_shapes_v1.txt
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_dma$i.log 2>&1; el "bench (dma) exit $?"; tail -1 $O/${P}_bench_dma$i.log | cut -c1-200
SDMI_ATTN_V1=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_v1$i.log 2>&1; el "bench (v1) exit $?"; tail -1 $O/${P}_bench_v1$i.log | cut -c1-200
done
el done
