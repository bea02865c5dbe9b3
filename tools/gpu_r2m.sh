#!/bin/bash
# attention: software-pipelined kernel vs the plain LDS-DMA ring (SDMI_ATTN_PIPE_MIN=1000000) vs register-staged (SDMI_ATTN_V1=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-m}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "attention or attn" -s > $O/${P}_attn.log 2>&1; el "attention kernel tests exit $? : $(tail -1 $O/${P}_attn.log)"
grep -E "^FAILED|^ERROR" $O/${P}_attn.log | head -30; grep -E "attention d" $O/${P}_attn.log | sed 's/^[.F]*//' | cut -c1-150 | head -40
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_clip_gpu.py tests/test_vae_gpu.py tests/test_pipeline_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet/clip/vae/pipeline tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -E "^FAILED|^ERROR" $O/${P}_unet.log | head; grep -E "\[unet .*max-abs" $O/${P}_unet.log | sed 's/^[.F]*//' | head -24
timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_pipe.txt 2>&1; el "prof_shapes (pipe) exit $?"; grep -E "^total|^attn" $O/${P}_shapes_pipe.txt
SDMI_ATTN_PIPE_MIN=1000000 timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_dma.txt 2>&1; el "prof_shapes (dma) exit $?"; grep -E "^total|^attn" $O/${P}_shapes_dma.txt
SDMI_ATTN_V1=1 timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_v1.txt 2>&1; el "prof_shapes (v1) exit $?"; grep -E "^total|^attn" $O/${P}_shapes_v1.txt
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_pipe$i.log 2>&1; el "bench (pipe) exit $?"; tail -1 $O/${P}_bench_pipe$i.log | cut -c1-200
SDMI_ATTN_PIPE_MIN=1000000 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_dma$i.log 2>&1; el "bench (dma) exit $?"; tail -1 $O/${P}_bench_dma$i.log | cut -c1-200
done
el done
