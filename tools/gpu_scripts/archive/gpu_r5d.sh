#!/bin/bash
# Round-5 GPU pass D (early evidence of the committed tree): the driver's sequence (pytest -m gpu, smoke, bench with the driver's
# arguments) + rocprofv3 --kernel-trace --stats of the bench command, all on ONE box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5d}
O=$PWD/gpurun_out; mkdir -p $O/${P}_benchprof
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/${P}_pytest.log 2>&1; el "pytest exit $? : $(tail -1 $O/${P}_pytest.log)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; el "smoke exit $?"; grep smoke: $O/${P}_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log > $O/${P}_bench.json; cut -c1-400 $O/${P}_bench.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${P}_benchprof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_benchprof.log 2>&1; el "prof exit $?"
python - "$P" <<'PY'
import sqlite3, glob, sys
P = sys.argv[1]
for f in glob.glob(f'gpurun_out/{P}_benchprof/*_results.db'):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open(f'gpurun_out/{P}_kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:70]:
            out.write(f'{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n')
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open(f'gpurun_out/{P}_kernel_stats.txt').read()[:1800])
PY
rm -rf $O/${P}_benchprof
el done
