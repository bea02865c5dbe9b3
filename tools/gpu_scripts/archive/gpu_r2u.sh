#!/bin/bash
# last short check of round 2: the whole GPU suite (-x) and one per-shape table (LayerNorm per width class; SDMI_LN_SLOTS=0 = before)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-u}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests -q -p no:cacheprovider -x -m gpu > $O/${P}_tests.log 2>&1; el "gpu suite exit $? : $(tail -1 $O/${P}_tests.log)"
timeout 100 python tools/prof_shapes.py > $O/${P}_shapes_new.txt 2>&1; el "prof_shapes new: $(grep -E '^total|^layernorm' $O/${P}_shapes_new.txt | tr '\n' ' ')"
SDMI_LN_SLOTS=0 timeout 100 python tools/prof_shapes.py > $O/${P}_shapes_ln5.txt 2>&1; el "prof_shapes LN 5-slot: $(grep -E '^total|^layernorm' $O/${P}_shapes_ln5.txt | tr '\n' ' ')"
el done
