#!/bin/bash
# Round-2 GPU pass D: kernel tests (incl. halo-staged conv), re-tune in situ, parity + bench + profile with the new table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-d}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider > $O/${P}_kernels.log 2>&1; el "kernel tests exit $? : $(tail -1 $O/${P}_kernels.log)"
grep -E "^FAILED|^ERROR" $O/${P}_kernels.log | head -30
timeout 900 python tools/tune.py --out $O/tune_gfx950.txt --dump $O/tune_dump.txt > $O/${P}_tune.log 2>&1; el "tune exit $? : $(tail -1 $O/${P}_tune.log)"
export SDMI_TUNE_FILE=$O/tune_gfx950.txt
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py tests/test_vae_gpu.py tests/test_clip_gpu.py tests/test_sampler_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet/pipeline/vae/clip/sampler tests (new table) exit $? : $(tail -1 $O/${P}_unet.log)"
grep -E "^FAILED|^ERROR" $O/${P}_unet.log | head; grep -E "\[unet .*max-abs|\[pipeline" $O/${P}_unet.log | sed 's/^[.F]*//' | head -24
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log | cut -c1-330
timeout 600 python tools/prof_shapes.py > $O/${P}_shapes.txt 2>&1; el "prof_shapes exit $?"; head -45 $O/${P}_shapes.txt
mkdir -p $O/${P}_prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${P}_prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_prof.log 2>&1; el "rocprof exit $?"
P=$P python - <<'PY'
import sqlite3, glob, os
P=os.environ['P']
for f in glob.glob(f'gpurun_out/{P}_prof/**/*_results.db', recursive=True):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open(f'gpurun_out/{P}_kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline\n(2 images = 102 UNet calls + 2 first-stage decodes + 12 UNet calls / 6 decodes of the latency probes)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:70]:
            out.write(f"{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n")
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open(f'gpurun_out/{P}_kernel_stats.txt').read()[:2600])
    os.remove(f)
PY
el done
