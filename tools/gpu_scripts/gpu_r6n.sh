#!/bin/bash
# Round 6, pass N: look for a SLOW-class box (6.4 - 6.6 ms per UNet call) and, on one, validate the round-6 tree:
# the knobs' A/Bs, a bench line, the GPU suite.  On a fast-class box: the latency line only (about 40 s of box time).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6n}
O=$PWD/gpurun_out; mkdir -p $O
MS=$(timeout 300 python tools/unet_latency.py probe 20 1 2>&1 | grep round | awk '{print $(NF-4)}')
echo "probe: $MS ms per UNet call" | tee -a $O/${P}_boxes.log
python - "$MS" <<'PY' || exit 0
import sys
sys.exit(0 if float(sys.argv[1] or 0) > 5.9 else 1)
PY
echo "SLOW-class box" | tee -a $O/${P}_boxes.log
timeout 900 python tools/unet_ab.py base SDMI_ATTN_KVS=0 SDMI_CONV_IN_STATS=0 SDMI_REPLAY=0 --rounds 3 > $O/${P}_slow_ab.log 2>&1; tail -4 $O/${P}_slow_ab.log
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/${P}_slow_bench.log 2>&1; tail -1 $O/${P}_slow_bench.log > $O/${P}_slow_bench.json; cut -c1-200 $O/${P}_slow_bench.json
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/${P}_slow_pytest.log 2>&1; echo "pytest exit $? : $(tail -1 $O/${P}_slow_pytest.log)"
