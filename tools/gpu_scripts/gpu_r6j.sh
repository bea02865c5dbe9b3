#!/bin/bash
# Round 6, pass J: conv_in emits the GroupNorm statistics of its output (two statistics launches fewer): goldens + A/B; key-split attention default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6j}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet goldens exit $? : $(tail -1 $O/${P}_unet.log)"
grep -a "^.\?\[unet" $O/${P}_unet.log | grep -v "held to" | cut -c1-120
timeout 900 python tools/unet_ab.py SDMI_CONV_IN_STATS=0 base SDMI_ATTN_KVS=0 --rounds 5 > $O/${P}_ab.log 2>&1; el "ab exit $?"; tail -3 $O/${P}_ab.log
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes.log 2>&1; grep "^total\|^groupnorm\|gn_stats\|conv_in" $O/${P}_shapes.log | cut -c1-150
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/${P}_pytest.log 2>&1; el "pytest exit $? : $(tail -1 $O/${P}_pytest.log)"; grep "FAILED\|^E " $O/${P}_pytest.log | head
el done
