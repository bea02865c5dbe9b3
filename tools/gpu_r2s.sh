#!/bin/bash
# L2 warm-up of the 1x1 GEMMs (SDMI_IGEMM_PREFETCH = k-tiles touched up front): tests + same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-s}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
SDMI_IGEMM_PREFETCH=64 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py -q -p no:cacheprovider -x -k "igemm or conv or halo or unet" > $O/${P}_kernels.log 2>&1; el "tests with prefetch exit $? : $(tail -1 $O/${P}_kernels.log)"
for pf in 0 8 32 64; do
SDMI_IGEMM_PREFETCH=$pf timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_$pf.txt 2>&1; el "prof_shapes prefetch $pf: $(grep ^total $O/${P}_shapes_$pf.txt)"
done
for i in 1 2; do
for pf in 0 8 64; do
SDMI_IGEMM_PREFETCH=$pf timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_$pf$i.log 2>&1; el "bench prefetch $pf: $(tail -1 $O/${P}_bench_$pf$i.log | cut -c60-110)"
done
done
el done
