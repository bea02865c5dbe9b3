#!/bin/bash
# Round-4 GPU pass D: unfused split-K slabs in the MFMA register order (16-byte write-through slab stores + lane transposes in
# splitk_reduce_tiled_kernel, SDMI_SLAB_TILED): kernel tests (bit-identity with the row-major slabs), UNet goldens, same-box A/B,
# and its interplay with the GroupNorm-applying reduction (which keeps row-major slabs at the 15 conv1 sites).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-d}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "splitk or register_order or statistics or scatter or split16 or halo or reduce_applies" > $O/${P}_kern.log 2>&1; el "split-K kernel tests exit $? : $(tail -1 $O/${P}_kern.log)"
grep -h "^FAILED\|Error" $O/${P}_kern.log | cut -c1-200 | head -20
timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "\[unet \|headroom\|^FAILED" $O/${P}_unet.log | sed 's/^\.*//' | cut -c1-170 | head -24
for r in 1 2 3; do
  SDMI_SLAB_TILED=0 timeout 300 python tools/unet_latency.py "row-major slabs (pass C state)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  timeout 300 python tools/unet_latency.py "register-order slabs" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_REDUCE_GN=0 timeout 300 python tools/unet_latency.py "register-order slabs, REDUCE_GN=0" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B"; cat $O/${P}_ab.txt
SDMI_SLAB_TILED=0 SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_rowmajor.txt 2>&1; el "per-shape table row-major exit $?"
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_tiled.txt 2>&1; el "per-shape table tiled exit $?"
python - <<'PY'
import re
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r'(\S+)\s+n=\s*(\d+) total\s+([\d.]+) us', l)
        if m: d[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load('gpurun_out/d_shapes_rowmajor.txt'), load('gpurun_out/d_shapes_tiled.txt')
rows = sorted(((b[k][1] - a[k][1], k) for k in a if k in b))
print('largest per-class changes, register-order - row-major (us per UNet call):')
for d, k in rows[:14] + rows[-6:]:
    print(f'  {d:+8.1f}  {k:58s} n={a[k][0]:3d} row-major {a[k][1]:8.1f}')
print('sum', round(sum(d for d, _ in rows), 1))
PY
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/${P}_all.log 2>&1; el "whole GPU suite exit $? : $(tail -1 $O/${P}_all.log)"
grep -h "^FAILED\|^ERROR" $O/${P}_all.log | head
el done
