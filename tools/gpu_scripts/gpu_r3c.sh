#!/bin/bash
# Round-3 GPU pass C: bisect of the GroupNorm-folding conv failures of pass B (two fp32 sources with small channel counts), the
# split-fp16 GEMM family (kernel tests + same-box A/B against the K-concatenated formulation), UNet goldens per configuration.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "gn_fold" > $O/c_kern.log 2>&1; el "gn_fold kernel tests exit $? : $(tail -1 $O/c_kern.log)"
grep -h "^\[conv3 gn-fold\|^FAILED\|^PASSED" $O/c_kern.log | cut -c1-150
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "split16 or split_fp16" > $O/c_s16.log 2>&1; el "split16 kernel tests exit $? : $(tail -1 $O/c_s16.log)"
grep -h "FAILED\|Error" $O/c_s16.log | head
for f in 0 1; do
  SDMI_FUSE_GN_CONV=$f timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -k "golden" > $O/c_unet_f$f.log 2>&1; el "unet goldens FUSE_GN_CONV=$f exit $? : $(tail -1 $O/c_unet_f$f.log)"
  grep -h "^\[unet " $O/c_unet_f$f.log | cut -c1-130
done
for r in 1 2; do
  SDMI_FUSE_GN_CONV=0 SDMI_SPLIT16_KERNEL=0 timeout 300 python tools/unet_latency.py "fold0 split16-0" 20 2 2>/dev/null | grep round >> $O/c_ab.txt
  SDMI_FUSE_GN_CONV=0 SDMI_SPLIT16_KERNEL=1 timeout 300 python tools/unet_latency.py "fold0 split16-1" 20 2 2>/dev/null | grep round >> $O/c_ab.txt
done
el "A/B split16"; cat $O/c_ab.txt
el done
