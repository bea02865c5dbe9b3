#!/bin/bash
# Round-4 GPU pass H: XCD-affine block numbering of GroupNorm-apply (SDMI_GN_XCD): kernel tests under the knob, UNet bit-identity, same-box A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-h}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
SDMI_GN_XCD=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "groupnorm" > $O/${P}_kern.log 2>&1; el "groupnorm kernel tests (GN_XCD=1) exit $? : $(tail -1 $O/${P}_kern.log)"
SDMI_GN_XCD=1 timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet goldens (GN_XCD=1) exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "headroom" $O/${P}_unet.log | sed 's/^\.*//' | cut -c1-120
for r in 1 2 3; do
  timeout 300 python tools/unet_latency.py "GroupNorm-apply blocks round-robin over XCDs" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_GN_XCD=1 timeout 300 python tools/unet_latency.py "GroupNorm-apply blocks XCD-affine" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B"; cat $O/${P}_ab.txt
el done
