#!/bin/bash
# tile order inside an XCD's range (SDMI_TILE_ORDER: 0 auto, 1 M fastest = old, 2 N fastest): tests + same-box A/B + per-shape tables
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-r}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "igemm or conv or halo" > $O/${P}_kernels.log 2>&1; el "igemm kernel tests exit $? : $(tail -1 $O/${P}_kernels.log)"
for ord in 0 1 2; do
SDMI_TILE_ORDER=$ord timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_$ord.txt 2>&1; el "prof_shapes order $ord: $(grep ^total $O/${P}_shapes_$ord.txt)"
done
for i in 1 2; do
for ord in 0 1 2; do
SDMI_TILE_ORDER=$ord timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_$ord$i.log 2>&1; el "bench order $ord: $(tail -1 $O/${P}_bench_$ord$i.log | cut -c60-110)"
done
done
el done
