#!/bin/bash
# Round-3 GPU pass V (closing): the committed tree -- pytest -m gpu -x, smoke, the three bench workloads on ONE box, LayerNorm fold
# threshold A/B (128 rows: the 8x8 mid block too).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/v_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/v_pytest.log)"
timeout 600 python __graft_entry__.py --smoke > $O/v_smoke.log 2>&1; el "smoke exit $? : $(grep -h smoke $O/v_smoke.log | head -3 | tr '\n' '|' | cut -c1-260)"
timeout 900 python bench.py > $O/v_bench.log 2>&1; el "bench exit $? : $(tail -1 $O/v_bench.log | cut -c1-150)"
timeout 600 python bench.py --workload txt2img768 --no-cpu-baseline > $O/v_bench768.log 2>&1; el "bench 768 exit $? : $(tail -1 $O/v_bench768.log | cut -c1-130)"
timeout 600 python bench.py --workload img2img512 --no-cpu-baseline > $O/v_benchi2i.log 2>&1; el "bench img2img exit $? : $(tail -1 $O/v_benchi2i.log | cut -c1-130)"
for r in 1 2; do
  timeout 300 python tools/unet_latency.py "LN fold from 512 rows (default)" 20 2 2>/dev/null | grep round >> $O/v_ab.txt
  SDMI_LN_FOLD_MIN_ROWS=128 timeout 300 python tools/unet_latency.py "LN fold from 128 rows" 20 2 2>/dev/null | grep round >> $O/v_ab.txt
done
el "A/B"; cat $O/v_ab.txt
el done
