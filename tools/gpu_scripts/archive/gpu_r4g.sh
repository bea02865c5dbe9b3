#!/bin/bash
# Round-4 GPU pass G: non-temporal weight-tile loads in the GEMM kernels (-DSDMI_W_AUX=2 build, libsdmi_wnt.so): same-box A/B against the
# default policy, per-class table of both, the 768 workload (more row tiles re-read every weight tile there).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-g}
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
SDMI_LIB_PATH=$L/libsdmi_wnt.so timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -k "golden" > $O/${P}_unet.log 2>&1; el "unet goldens (nt weights) exit $? : $(tail -1 $O/${P}_unet.log)"
for r in 1 2 3; do
  timeout 300 python tools/unet_latency.py "default weight-load policy" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_LIB_PATH=$L/libsdmi_wnt.so timeout 300 python tools/unet_latency.py "nt weight loads" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B"; cat $O/${P}_ab.txt
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_default.txt 2>&1; el "per-shape table default exit $?"
SDMI_LIB_PATH=$L/libsdmi_wnt.so SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_wnt.txt 2>&1; el "per-shape table nt exit $?"
P=$P python - <<'PY'
import re, os
P = os.environ['P']
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r'(\S+)\s+n=\s*(\d+) total\s+([\d.]+) us', l)
        if m: d[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load(f'gpurun_out/{P}_shapes_default.txt'), load(f'gpurun_out/{P}_shapes_wnt.txt')
rows = sorted(((b[k][1] - a[k][1], k) for k in a if k in b))
print('per-class changes, nt - default (us per UNet call), all classes with |change| >= 2:')
for d, k in rows:
    if abs(d) >= 2: print(f'  {d:+8.1f}  {k:58s} n={a[k][0]:3d} default {a[k][1]:8.1f}')
print('sum', round(sum(d for d, _ in rows), 1))
PY
for w in txt2img768; do
  timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_default_$w.log 2>&1; el "bench $w default exit $? : $(tail -1 $O/${P}_bench_default_$w.log | cut -c60-130)"
  SDMI_LIB_PATH=$L/libsdmi_wnt.so timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_wnt_$w.log 2>&1; el "bench $w nt exit $? : $(tail -1 $O/${P}_bench_wnt_$w.log | cut -c60-130)"
done
el done
