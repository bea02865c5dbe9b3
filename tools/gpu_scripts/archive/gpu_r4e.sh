#!/bin/bash
# Round-4 GPU pass E: the GroupNorm-applying split-K reduction reading register-order slabs (every split site tiled now): kernel tests,
# UNet goldens + bit-identity tests, same-box A/B of the four combinations of SDMI_SLAB_TILED / SDMI_REDUCE_GN, per-shape table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-e}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "splitk or register_order or statistics or scatter or split16 or halo or reduce_applies" > $O/${P}_kern.log 2>&1; el "split-K kernel tests exit $? : $(tail -1 $O/${P}_kern.log)"
grep -h "^FAILED\|Error" $O/${P}_kern.log | cut -c1-200 | head -20
timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "headroom\|\[reduce+gn\|^FAILED" $O/${P}_unet.log | sed 's/^\.*//' | cut -c1-170 | head -24
for r in 1 2 3; do
  SDMI_SLAB_TILED=0 timeout 300 python tools/unet_latency.py "row-major slabs, GN in the reduction (pass C)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_REDUCE_GN=0 timeout 300 python tools/unet_latency.py "register-order slabs, GN-apply launches" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  timeout 300 python tools/unet_latency.py "register-order slabs, GN in the reduction" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B"; cat $O/${P}_ab.txt
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes.txt 2>&1; el "per-shape table exit $?"; grep -v amdgpu $O/${P}_shapes.txt | head -12
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/${P}_all.log 2>&1; el "whole GPU suite exit $? : $(tail -1 $O/${P}_all.log)"
grep -h "^FAILED\|^ERROR" $O/${P}_all.log | head
el done
