#!/bin/bash
# Round-3 GPU pass T: the committed tree with the merged tuning table: pytest -m gpu -x, default bench, rocprofv3 kernel stats of the
# bench command, per-shape table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O/t_benchprof
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/t_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/t_pytest.log)"
timeout 900 python bench.py > $O/t_bench.log 2>&1; el "bench exit $? : $(tail -1 $O/t_bench.log | cut -c1-160)"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/t_benchprof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/t_prof.log 2>&1; el "rocprofv3 exit $?"
python - <<'PY'
import sqlite3, glob
for f in glob.glob('gpurun_out/t_benchprof/**/*_results.db', recursive=True):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open('gpurun_out/t_kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline   (2 images = 102 UNet calls + 2 first-stage decodes + the latency probes)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:70]:
            out.write(f'{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n')
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open('gpurun_out/t_kernel_stats.txt').read()[:900])
PY
find $O/t_benchprof -type f ! -name '*.txt' ! -name '*.log' -delete; find $O -type d -empty -delete
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/t_shapes.txt 2>&1; el "prof_shapes exit $?"; grep -v amdgpu $O/t_shapes.txt | head -4
el done
