#!/bin/bash
# Round-4 GPU pass B: the split-K reduction that applies the GroupNorm with the two-launch path's statistics (bit-identity on whole
# UNet calls), then same-box A/Bs that decide two value-neutral defaults: write-through (sc1) 16-byte fp32 output stores
# (libsdmi_wt1.so, -DSDMI_WT_STORES=1) and GroupNorm-apply with 4 quads per thread on the 64x64 level -- UNet latency, the bench
# line (first-stage decode included) and per-class tables under both libraries.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-b}
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -s -k "reduce_applies_groupnorm or test_groupnorm" > $O/${P}_kern_rgn.log 2>&1; el "reduce+gn / groupnorm kernel tests exit $? : $(tail -1 $O/${P}_kern_rgn.log)"
grep -h "^\[reduce+gn\|^FAILED\|Error" $O/${P}_kern_rgn.log | cut -c1-200 | head -20
timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "\[unet \|headroom\|\[reduce+gn\|^FAILED" $O/${P}_unet.log | sed 's/^\.*//' | cut -c1-200 | head -40
for r in 1 2 3; do
  timeout 300 python tools/unet_latency.py "plain stores" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_LIB_PATH=$L/libsdmi_wt1.so timeout 300 python tools/unet_latency.py "sc1 fp32 16-byte stores (wt1)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_GN_APPLY_U4_QUADS=300000 timeout 300 python tools/unet_latency.py "plain + gn_apply U4 at 64x64" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_LIB_PATH=$L/libsdmi_wt1.so SDMI_GN_APPLY_U4_QUADS=300000 timeout 300 python tools/unet_latency.py "wt1 + gn_apply U4 at 64x64" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B"; cat $O/${P}_ab.txt
for w in txt2img512 txt2img768 img2img512; do
  timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_plain_$w.log 2>&1; el "bench $w plain exit $?"; tail -1 $O/${P}_bench_plain_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('value','ms_per_step','unet_ms_per_call','vae_decode_ms') if k in d})"
  SDMI_LIB_PATH=$L/libsdmi_wt1.so SDMI_GN_APPLY_U4_QUADS=300000 timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_wt1_$w.log 2>&1; el "bench $w wt1+U4 exit $?"; tail -1 $O/${P}_bench_wt1_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('value','ms_per_step','unet_ms_per_call','vae_decode_ms') if k in d})"
done
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_plain.txt 2>&1; el "per-shape table plain exit $?"
SDMI_LIB_PATH=$L/libsdmi_wt1.so SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_wt1.txt 2>&1; el "per-shape table wt1 exit $?"
python - <<'E'
import re, os
O = os.environ.get('O', 'gpurun_out')
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r'(\S+)\s+n=\s*(\d+) total\s+([\d.]+) us', l)
        if m: d[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return d
import sys
P = sys.argv[1] if len(sys.argv) > 1 else 'b'
a, b = load(f'gpurun_out/{P}_shapes_plain.txt'), load(f'gpurun_out/{P}_shapes_wt1.txt')
rows = sorted(((b[k][1] - a[k][1], k) for k in a if k in b))
print('largest per-class changes, wt1 - plain (us per UNet call):')
for d, k in rows[:12] + rows[-8:]:
    print(f'  {d:+8.1f}  {k:58s} n={a[k][0]:3d} plain {a[k][1]:8.1f}')
print('sum', sum(d for d, _ in rows))
E
# speculative (if the A/Bs above favour wt1 + U4): re-tune the bench workload's shapes on that library, A/B the tables, parity with the new one
export SDMI_LIB_PATH=$L/libsdmi_wt1.so SDMI_GN_APPLY_U4_QUADS=300000
cp $L/tune_gfx950.txt $O/${P}_tune.txt
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 600 python tools/tune.py --workloads unet64 --rounds 72 --reps 4 --out $O/${P}_tune.txt --dump $O/${P}_tune_dump.txt > $O/${P}_tune.log 2>&1; el "tune unet64 exit $? : $(tail -1 $O/${P}_tune.log)"
for r in 1 2; do
  timeout 300 python tools/unet_latency.py "wt1+U4, committed table" 20 2 2>/dev/null | grep round >> $O/${P}_ab2.txt
  SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 300 python tools/unet_latency.py "wt1+U4, re-tuned table" 20 2 2>/dev/null | grep round >> $O/${P}_ab2.txt
done
el "A/B tables"; cat $O/${P}_ab2.txt
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet2.log 2>&1; el "unet goldens (re-tuned table) exit $? : $(tail -1 $O/${P}_unet2.log)"
grep -h "\[unet \|headroom\|^FAILED" $O/${P}_unet2.log | sed 's/^\.*//' | cut -c1-170 | head -20
diff <(grep -v "^#" $L/tune_gfx950.txt | cut -d" " -f1-10 | sort) <(grep -v "^#" $O/${P}_tune.txt | cut -d" " -f1-10 | sort) | grep -c "^>" | xargs echo "table rows changed:"
el done
