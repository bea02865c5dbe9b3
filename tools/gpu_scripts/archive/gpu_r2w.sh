#!/bin/bash
# Round-2 late changes, same box: 16-byte epilogues (q / k scatter + unsplit plain mode; SDMI_EPI_VEC), LDS-staged small_linear
# (SDMI_SMALL_LDS), 4-pixels-per-wave conv_out (SDMI_CONV_OUT4): bit-identity tests, the whole GPU suite, interleaved bench A/B,
# per-shape tables, per-workgroup phase timing (instrumented library, when present).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-w}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests -q -p no:cacheprovider -m gpu -k "16_byte or conv_out_4 or small_linear_lds" > $O/${P}_bitident.log 2>&1; el "bit-identity tests exit $? : $(tail -1 $O/${P}_bitident.log)"
timeout 900 python -m pytest tests -q -p no:cacheprovider -x -m gpu > $O/${P}_tests.log 2>&1; el "gpu suite exit $? : $(tail -1 $O/${P}_tests.log)"
OFF="SDMI_EPI_VEC=0 SDMI_SMALL_LDS=0 SDMI_CONV_OUT4=0"
env $OFF timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_off.txt 2>&1; el "prof_shapes all off: $(grep ^total $O/${P}_shapes_off.txt)"
timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_on.txt 2>&1; el "prof_shapes all on: $(grep ^total $O/${P}_shapes_on.txt)"
for i in 1 2; do
for v in "A=1" "$OFF" "SDMI_EPI_VEC=0" "SDMI_SMALL_LDS=0 SDMI_CONV_OUT4=0"; do
env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench.tmp 2>&1; el "bench [$v]: $(tail -1 $O/${P}_bench.tmp | cut -c60-110)"
cat $O/${P}_bench.tmp >> $O/${P}_bench_all.log
done
done
if [ -f stable-diffusion_amd/libsdmi_timing.so ]; then
for v in 1 0; do
SDMI_EPI_VEC=$v SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_timing.so timeout 300 python tools/igemm_timing.py $O/${P}_timing_$v.txt > $O/${P}_timing_$v.summary 2>&1; el "phase timing vec $v: $(head -2 $O/${P}_timing_$v.summary | tail -1)"
done
fi
el done
