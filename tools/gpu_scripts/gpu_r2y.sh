#!/bin/bash
# Round-2 closing evidence on one box: measured HBM traffic (bench.py --traffic-pass), the default bench line with roofline + cpu_baseline,
# the other two workloads, torchrun N=1, rocprofv3 kernel stats of the bench command.  (Tests / smoke: gpu_r2x.sh.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O/y_benchprof; P=${1:-y}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python bench.py > $O/${P}_bench.log 2>&1; el "bench exit $? : $(tail -1 $O/${P}_bench.log | cut -c1-120)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_torchrun1.log 2>&1; el "torchrun N=1 exit $? : $(tail -1 $O/${P}_torchrun1.log | cut -c60-110)"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/y_benchprof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_prof.log 2>&1; el "rocprofv3 exit $?"
python - <<'PY'
import sqlite3, glob
for f in glob.glob('gpurun_out/y_benchprof/**/*_results.db', recursive=True):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open('gpurun_out/y_kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline   (2 images = 102 UNet calls + 2 first-stage decodes + the latency probes)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:60]:
            out.write(f'{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n')
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open('gpurun_out/y_kernel_stats.txt').read()[:1800])
PY
timeout 600 python bench.py --traffic-pass --traffic-out $O/${P}_traffic.json > $O/${P}_traffic.log 2>&1; el "traffic pass exit $?"
find $O/y_benchprof -type f ! -name '*.txt' ! -name '*.log' -delete; find $O -type d -empty -delete
timeout 600 python bench.py --workload txt2img768 --no-cpu-baseline > $O/${P}_bench768.log 2>&1; el "bench 768 exit $? : $(tail -1 $O/${P}_bench768.log | cut -c1-120)"
timeout 600 python bench.py --workload img2img512 --no-cpu-baseline > $O/${P}_benchi2i.log 2>&1; el "bench img2img exit $? : $(tail -1 $O/${P}_benchi2i.log | cut -c1-120)"
el done
