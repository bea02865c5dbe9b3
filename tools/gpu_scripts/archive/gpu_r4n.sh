#!/bin/bash
# Round-4 GPU pass N: does the workgroup count matter for the <= 1024-query attention launches (d = 80 / 160 self-attention at 1024 / 256 queries,
# every cross-attention: the launches a KV-split would spread over more CUs)?  SDMI_ATTN_NW_LE1K = waves per workgroup: 4 (default) -> 2 doubles the
# workgroups of those launches (d160 self: 32 -> 64, cross d160: 32 -> 64), 8 halves them.  Same-box A/B + per-class times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-n}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for r in 1 2; do
  for nw in 4 2 8; do
    SDMI_ATTN_NW_LE1K=$nw timeout 300 python tools/unet_latency.py "<= 1024 queries: $nw waves per workgroup" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  done
done
el "A/B"; cat $O/${P}_ab.txt
for nw in 4 2 8; do
  SDMI_ATTN_NW_LE1K=$nw SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py 2>/dev/null | grep "^attn_" | sed "s/^/NW=$nw  /"
done
el done
