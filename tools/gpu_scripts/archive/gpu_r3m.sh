#!/bin/bash
# Round-3 GPU pass M: per-case parity numbers of the committed build + table (-s), runtime knobs: kernel arguments in device memory
# (HIP_FORCE_DEV_KERNARG), static wave priority in the 8-wave attention kernel (SDMI_ATTN_PRIO).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_unet_gpu.py -q -s -p no:cacheprovider -k "golden or headroom or noise_floor or batch_rows or TUNE_DISABLE or tune" > $O/m_unet.log 2>&1; el "unet parity (-s) exit $? : $(tail -1 $O/m_unet.log)"
grep -h "^\[unet \|^\[noise" $O/m_unet.log | cut -c1-200
for r in 1 2; do
  HIP_FORCE_DEV_KERNARG=0 timeout 300 python tools/unet_latency.py "HIP_FORCE_DEV_KERNARG=0" 20 2 2>/dev/null | grep round >> $O/m_ab.txt
  HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/unet_latency.py "HIP_FORCE_DEV_KERNARG=1" 20 2 2>/dev/null | grep round >> $O/m_ab.txt
  timeout 300 python tools/unet_latency.py "default env" 20 2 2>/dev/null | grep round >> $O/m_ab.txt
  SDMI_ATTN_PRIO=1 timeout 300 python tools/unet_latency.py "SDMI_ATTN_PRIO=1" 20 2 2>/dev/null | grep round >> $O/m_ab.txt
done
el "A/B"; cat $O/m_ab.txt
SDMI_ATTN_PRIO=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "attention" > $O/m_attn.log 2>&1; el "attention tests with PRIO=1 exit $? : $(tail -1 $O/m_attn.log)"
el done
