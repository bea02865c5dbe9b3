#!/bin/bash
# Value-neutral launch-level changes (loads hoisted in front of waits / barriers, fp16 copies from the producers, timestep table):
# bit-identity tests, whole GPU suite, smoke, same-box A/B against the previous build (SDMI_LIB_PATH=.../libsdmi_prev.so, when present).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-x}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests -q -p no:cacheprovider -m gpu -k "16_byte or conv_out_4 or small_linear_lds or timestep_table" > $O/${P}_bitident.log 2>&1; el "bit-identity tests exit $? : $(tail -1 $O/${P}_bitident.log)"
timeout 900 python -m pytest tests -q -p no:cacheprovider -x -m gpu > $O/${P}_tests.log 2>&1; el "gpu suite exit $? : $(tail -1 $O/${P}_tests.log)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; el "smoke exit $?"; grep smoke: $O/${P}_smoke.log
PREV=$PWD/stable-diffusion_amd/libsdmi_prev.so
timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_new.txt 2>&1; el "prof_shapes new: $(grep ^total $O/${P}_shapes_new.txt)"
[ -f $PREV ] && { SDMI_LIB_PATH=$PREV timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_prev.txt 2>&1; el "prof_shapes prev: $(grep ^total $O/${P}_shapes_prev.txt)"; }
for i in 1 2; do
for v in "A=1" "SDMI_LIB_PATH=$PREV"; do
[ "$v" = "SDMI_LIB_PATH=$PREV" ] && [ ! -f $PREV ] && continue
env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench.tmp 2>&1; el "bench [${v##*/}]: $(tail -1 $O/${P}_bench.tmp | cut -c60-110)"
cat $O/${P}_bench.tmp >> $O/${P}_bench_all.log
done
done
el done
