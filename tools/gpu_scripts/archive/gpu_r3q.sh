#!/bin/bash
# Round-3 GPU pass Q: runtime / driver environment sweep on one box (does anything outside the library move the ~5 us launch cost?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
run() { label="$1"; shift; env "$@" timeout 300 python tools/unet_latency.py "$label" 20 2 2>/dev/null | grep round >> $O/q_ab.txt; }
for r in 1 2; do
  run "default env" X=1
  run "GPU_MAX_HW_QUEUES=1" GPU_MAX_HW_QUEUES=1
  run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
  run "AMD_DIRECT_DISPATCH=0" AMD_DIRECT_DISPATCH=0
  run "ROC_ACTIVE_WAIT_TIMEOUT=1000" ROC_ACTIVE_WAIT_TIMEOUT=1000
  run "HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0" AMD_SERIALIZE_KERNEL=0
  run "LN fold from 8192 rows only" SDMI_LN_FOLD_MIN_ROWS=8192
  run "LN fold off" SDMI_LN_FOLD=0
done
el "A/B"; cat $O/q_ab.txt
rocm-smi --showclocks --showperflevel --showpower 2>/dev/null | grep -v "^=\|^$" | head -20
el done
