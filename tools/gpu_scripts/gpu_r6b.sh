#!/bin/bash
# Round 6, pass B: goldens with the split-fp16 context K / V projections and the floor-aware criteria; the hierarchical grid barrier in the
# GroupNorm-applying split-K reduction (parity + same-box A/B).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6b}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1500 python -m pytest tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider -s -k "golden or headroom or grid_barrier" > $O/${P}_unet.log 2>&1; el "unet exit $? : $(tail -1 $O/${P}_unet.log)"
grep "^.\?\[unet\|^.\?\[reduce" $O/${P}_unet.log | cut -c1-230
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -s -k "gelu_erf or grid_barrier or splitk" > $O/${P}_kern.log 2>&1; el "kernels exit $? : $(tail -1 $O/${P}_kern.log)"; grep "gelu_erf\]\|grid barrier" $O/${P}_kern.log | cut -c1-200
timeout 600 python tools/unet_ab.py SDMI_REDUCE_GN_XCD=0 SDMI_REDUCE_GN_XCD=1 --rounds 5 > $O/${P}_ab.log 2>&1; el "ab exit $?"; cat $O/${P}_ab.log | tail -3
SDMI_REDUCE_GN_XCD=1 SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_xcd.log 2>&1; grep "^total\|^splitk_reduce\|^groupnorm" $O/${P}_shapes_xcd.log
SDMI_REDUCE_GN_XCD=0 SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_base.log 2>&1; grep "^total\|^splitk_reduce\|^groupnorm" $O/${P}_shapes_base.log
el done
