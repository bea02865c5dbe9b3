#!/bin/bash
# Round-4 GPU pass P: threads per block of the register-order split-K reduction (SDMI_REDUCE_BLOCK 256 / 128 / 0 = auto): bit-identity tests under
# 128, same-box A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-p}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
SDMI_REDUCE_BLOCK=128 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "splitk or register_order or statistics or scatter or split16" > $O/${P}_kern.log 2>&1; el "split-K kernel tests (128-thread reduce) exit $? : $(tail -1 $O/${P}_kern.log)"
SDMI_REDUCE_BLOCK=128 timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s -k "golden or headroom or register_order" > $O/${P}_unet.log 2>&1; el "unet tests (128) exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "headroom\|^FAILED" $O/${P}_unet.log | sed 's/^\.*//' | cut -c1-150 | head
for r in 1 2 3; do
  for rb in 256 128 0; do
    SDMI_REDUCE_BLOCK=$rb timeout 300 python tools/unet_latency.py "reduce blocks of $rb threads (0 = auto)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  done
done
el "A/B"; cat $O/${P}_ab.txt
el done
