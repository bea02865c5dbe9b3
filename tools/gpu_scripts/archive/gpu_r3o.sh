#!/bin/bash
# Round-3 GPU pass O: ping-pong attention (attn_pp_kernel, SDMI_ATTN_PP=1): bit-identity + parity tests, same-box A/B, per-shape times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "attention" > $O/o_kern.log 2>&1; el "attention kernel tests exit $? : $(tail -1 $O/o_kern.log)"
grep -h "^FAILED\|Error\|assert" $O/o_kern.log | cut -c1-200 | head
for r in 1 2; do
  SDMI_ATTN_PP=0 timeout 300 python tools/unet_latency.py "attention lock-step (PP=0)" 20 2 2>/dev/null | grep round >> $O/o_ab.txt
  SDMI_ATTN_PP=1 timeout 300 python tools/unet_latency.py "attention ping-pong (PP=1)" 20 2 2>/dev/null | grep round >> $O/o_ab.txt
  SDMI_ATTN_PP=1 SDMI_ATTN_NW_LE1K=8 timeout 300 python tools/unet_latency.py "ping-pong, 8 waves also <= 1024 queries" 20 2 2>/dev/null | grep round >> $O/o_ab.txt
done
el "A/B"; cat $O/o_ab.txt
for pp in 0 1; do SDMI_ATTN_PP=$pp SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py 2>/dev/null | grep "attn_\|^total" | cut -c1-120 | sed "s/^/PP=$pp  /"; done
SDMI_ATTN_PP=1 timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -k "golden" > $O/o_unet.log 2>&1; el "unet goldens (PP=1) exit $? : $(tail -1 $O/o_unet.log)"
el done
