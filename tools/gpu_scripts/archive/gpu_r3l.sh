#!/bin/bash
# Round-3 GPU pass L (evidence on one box): re-tune of ALL workloads on the round-3 kernels, then with that table: the driver's own
# sequence (pytest -m gpu -x, smoke, default bench with roofline + cpu_baseline), the 768 / img2img workloads, rocprofv3 kernel
# stats of the bench command, measured HBM traffic (bench.py --traffic-pass), per-shape table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O/l_benchprof
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
cp $L/tune_gfx950.txt $O/l_tune.txt
SDMI_TUNE_FILE=$O/l_tune.txt timeout 900 python tools/tune.py --workloads unet96,unet32,unet64b4,unet64b6,unet64b8,vaedec64,vaedec96,vaeenc512,clip --rounds 72 --reps 3 --out $O/l_tune.txt --dump $O/l_tune_dump.txt > $O/l_tune.log 2>&1; el "tune (all but unet64) exit $? : $(tail -1 $O/l_tune.log)"
cp $O/l_tune.txt $L/tune_gfx950.txt          # (the table this pass is judged with; committed afterwards)
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/l_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/l_pytest.log)"
grep -h "^\[unet \|headroom\|^FAILED" $O/l_pytest.log | cut -c1-170 | head -40
timeout 600 python __graft_entry__.py --smoke > $O/l_smoke.log 2>&1; el "smoke exit $? : $(grep -h smoke $O/l_smoke.log | head -3 | tr '\n' '|' | cut -c1-300)"
timeout 900 python bench.py > $O/l_bench.log 2>&1; el "bench exit $? : $(tail -1 $O/l_bench.log | cut -c1-160)"
timeout 600 python bench.py --workload txt2img768 --no-cpu-baseline > $O/l_bench768.log 2>&1; el "bench 768 exit $? : $(tail -1 $O/l_bench768.log | cut -c1-140)"
timeout 600 python bench.py --workload img2img512 --no-cpu-baseline > $O/l_benchi2i.log 2>&1; el "bench img2img exit $? : $(tail -1 $O/l_benchi2i.log | cut -c1-140)"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/l_benchprof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/l_prof.log 2>&1; el "rocprofv3 exit $?"
python - <<'PY'
import sqlite3, glob
for f in glob.glob('gpurun_out/l_benchprof/**/*_results.db', recursive=True):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open('gpurun_out/l_kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline   (2 images = 102 UNet calls + 2 first-stage decodes + the latency probes)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:70]:
            out.write(f'{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n')
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open('gpurun_out/l_kernel_stats.txt').read()[:1500])
PY
timeout 600 python bench.py --traffic-pass --traffic-out $O/l_traffic.json > $O/l_traffic.log 2>&1; el "traffic pass exit $? : $(tail -1 $O/l_traffic.log | cut -c1-200)"
find $O/l_benchprof -type f ! -name '*.txt' ! -name '*.log' -delete; find $O -type d -empty -delete
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/l_shapes.txt 2>&1; el "prof_shapes exit $?"; grep -v amdgpu $O/l_shapes.txt | head -3
el done
