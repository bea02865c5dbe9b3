#!/bin/bash
# Round 6, pass H: (1) first-stage / text-encoder errors as measured (for the tolerances), (2) VERDICT r5 item 4b: distinct kernel instantiations per UNet
# call vs ms per call -- the committed table against heuristic tiles (SDMI_TUNE_DISABLE=1: fewer instantiations), this box's class from box_probe,
# (3) a bench line with the new fields.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6h}
O=$PWD/gpurun_out; mkdir -p $O/${P}_kt
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_clip_gpu.py tests/test_pipeline_gpu.py -q -m gpu -p no:cacheprovider -s > $O/${P}_vaeclip.log 2>&1; el "vae/clip/pipeline exit $? : $(tail -1 $O/${P}_vaeclip.log)"
grep -a "^.\?\[vae\|^.\?\[clip\|^.\?\[pipeline\|max-abs" $O/${P}_vaeclip.log | cut -c1-200 | head -40
for cfg in table heuristic; do
  if [ $cfg = heuristic ]; then export SDMI_TUNE_DISABLE=1; else unset SDMI_TUNE_DISABLE; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${P}_kt/$cfg -o kt -- python tools/prof_shapes.py > $O/${P}_kt_$cfg.log 2>&1
  timeout 300 python tools/unet_latency.py "$cfg" 20 3 2>&1 | grep round
done > $O/${P}_distinct.log 2>&1; unset SDMI_TUNE_DISABLE; el "distinct exit $?"
python - "$P" >> $O/${P}_distinct.log <<'PY'
import sqlite3, glob, sys
P = sys.argv[1]
for cfg in ('table', 'heuristic'):
    for f in glob.glob(f'gpurun_out/{P}_kt/{cfg}/**/*_results.db', recursive=True):
        con = sqlite3.connect(f)
        rows = con.execute("select name,total_calls from top_kernels").fetchall()
        unet = [(n, c) for n, c in rows if 'sdmi' in n and 'pack_' not in n and 'ln_fold_prep' not in n and 'cast_f16' not in n]
        print(f'{cfg}: {len(unet)} distinct sdmi kernel instantiations in the trace of tools/prof_shapes.py (model build + UNet calls)')
PY
cat $O/${P}_distinct.log
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log > $O/${P}_bench.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6h_bench.json'))
print({k: d.get(k) for k in ('value', 'unet_ms_per_call', 'unet_host_enqueue_ms_per_call', 'unet_host_loop_ms_per_call', 'vae_decode_ms')})
print('box', {k: d['box_probe'][k] for k in ('mixed_kernels_us', 'same_kernel_us', 'empty_launch_us')})
r = d['roofline']; print({k: r[k] for k in ('frac', 'frac_events', 'frac_scaled', 'avg_launch_ms', 'launches_per_unet_call')})
print(d.get('roofline_weight_stream'))
print(d.get('cpu_baseline'))
PY
rm -rf $O/${P}_kt
el done
