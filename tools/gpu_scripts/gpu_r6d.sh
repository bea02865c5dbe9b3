#!/bin/bash
# Round 6, pass D: the final precision allocation (last ResBlock 3-pass with its halo tile, stream 1x1 single-pass at downsample factors >= 4):
# the WHOLE -m gpu suite, smoke, then old-vs-new allocation timing on this box and a bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6d}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/${P}_pytest.log 2>&1; el "pytest exit $? : $(tail -1 $O/${P}_pytest.log)"
grep "^.\?\[unet\|FAILED\|Error" $O/${P}_pytest.log | cut -c1-220 | head -50
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; el "smoke exit $?"; grep smoke: $O/${P}_smoke.log
X=$PWD/stable-diffusion_amd/libsdmi_exp.so
for rep in 1 2; do
  SDMI_LIB_PATH=$X SDMI_PRECISE_LAST_RES=0 SDMI_PRECISE_1X1_MAX_DS=99 SDMI_PRECISE_KV=0 timeout 300 python tools/unet_latency.py "r5 allocation" 20 2 2>&1 | grep round
  SDMI_LIB_PATH=$X SDMI_PRECISE_LAST_RES=0 timeout 300 python tools/unet_latency.py "1x1 single-pass at ds>=4" 20 2 2>&1 | grep round
  timeout 300 python tools/unet_latency.py "product (r6 allocation)" 20 2 2>&1 | grep round
done > $O/${P}_lat.log 2>&1; el "latency exit $?"; cat $O/${P}_lat.log
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log > $O/${P}_bench.json; cut -c1-200 $O/${P}_bench.json
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes.log 2>&1; grep "^total\|K17280_s1\|M8192_N320_K8640" $O/${P}_shapes.log | cut -c1-150
el done
