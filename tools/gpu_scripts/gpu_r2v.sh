#!/bin/bash
# 16-byte GEMM epilogues (SDMI_EPI_VEC: 1 default, 0 = dword / short epilogues): bit-identity tests, the whole GPU suite, same-box A/B
# (interleaved bench runs + per-shape tables) and, when the instrumented library is there, the per-workgroup phase timing of both.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-v}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests -q -p no:cacheprovider -x -m gpu -k "16_byte" > $O/${P}_bitident.log 2>&1; el "bit-identity tests exit $? : $(tail -1 $O/${P}_bitident.log)"
timeout 900 python -m pytest tests -q -p no:cacheprovider -x -m gpu > $O/${P}_tests.log 2>&1; el "gpu suite exit $? : $(tail -1 $O/${P}_tests.log)"
for v in 1 0; do
SDMI_EPI_VEC=$v timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_$v.txt 2>&1; el "prof_shapes vec $v: $(grep ^total $O/${P}_shapes_$v.txt)"
done
for i in 1 2; do
for v in 1 0; do
SDMI_EPI_VEC=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_$v$i.log 2>&1; el "bench vec $v: $(tail -1 $O/${P}_bench_$v$i.log | cut -c60-110)"
done
done
if [ -f stable-diffusion_amd/libsdmi_timing.so ]; then
for v in 1 0; do
SDMI_EPI_VEC=$v SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_timing.so timeout 300 python tools/igemm_timing.py $O/${P}_timing_$v.txt > $O/${P}_timing_$v.summary 2>&1; el "phase timing vec $v: $(head -2 $O/${P}_timing_$v.summary | tail -1)"
done
fi
el done
