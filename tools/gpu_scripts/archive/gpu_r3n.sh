#!/bin/bash
# Round-3 GPU pass N: cross-attention with the to_q projection inside the kernel (csrc/attn_ctx.hip): kernel tests, UNet goldens,
# same-box A/B against the two launches (SDMI_ATTN_CTX_FUSED=0), workgroup-width knob, per-shape table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "attention_ctx" > $O/n_kern.log 2>&1; el "attn_ctx kernel tests exit $? : $(tail -1 $O/n_kern.log)"
grep -o "\[attn_ctx [^]]*\][^[]*" $O/n_kern.log | cut -c1-150 | head -40
grep -h "^FAILED\|Error" $O/n_kern.log | cut -c1-200 | head
timeout 600 python -m pytest tests/test_unet_gpu.py -q -s -p no:cacheprovider -k "golden or headroom" > $O/n_unet.log 2>&1; el "unet goldens exit $? : $(tail -1 $O/n_unet.log)"
grep -o "\[unet [^]]*\][^[]*" $O/n_unet.log | cut -c1-150 | head -20
for r in 1 2; do
  SDMI_ATTN_CTX_FUSED=0 timeout 300 python tools/unet_latency.py "to_q GEMM + attention (2 launches)" 20 2 2>/dev/null | grep round >> $O/n_ab.txt
  timeout 300 python tools/unet_latency.py "to_q inside the attention kernel" 20 2 2>/dev/null | grep round >> $O/n_ab.txt
  SDMI_ATTN_CTX_MAXD=80 timeout 300 python tools/unet_latency.py "fused, head dims <= 80" 20 2 2>/dev/null | grep round >> $O/n_ab.txt
  SDMI_ATTN_CTX_MAXD=40 timeout 300 python tools/unet_latency.py "fused, head dim 40 only" 20 2 2>/dev/null | grep round >> $O/n_ab.txt
done
el "A/B"; cat $O/n_ab.txt
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/n_shapes.txt 2>&1; el "prof_shapes exit $?"; grep -v amdgpu $O/n_shapes.txt | head -3; grep "attn" $O/n_shapes.txt | cut -c1-130
el done
