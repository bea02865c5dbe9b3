#!/bin/bash
# bench.py once plain (official numbers) and once under rocprofv3 --kernel-trace --stats (per-kernel summary)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O/benchprof
timeout 900 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 ${BENCH_ARGS} > $O/bench_full.log 2>&1; echo "bench exit $?"
tail -1 $O/bench_full.log | cut -c1-400
timeout 900 rocprofv3 --kernel-trace --stats -d $O/benchprof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/benchprof.log 2>&1; echo "prof exit $?"
python - <<'PY'
import sqlite3, glob
for f in glob.glob('gpurun_out/benchprof/*_results.db'):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open('gpurun_out/benchprof/kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 (2 images = 102 UNet calls + 2 VAE decodes)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:60]:
            out.write(f'{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n')
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open('gpurun_out/benchprof/kernel_stats.txt').read()[:6000])
PY
