#!/bin/bash
# Round-4 GPU pass C (evidence on one box, the committed defaults: write-through fp32 stores, GroupNorm-apply U4 at 64x64, GroupNorm in
# the split-K reduction): the driver's own sequence (pytest -m gpu -x, smoke, default bench with roofline + cpu_baseline), the 768 /
# img2img workloads, torchrun N = 1 (RCCL), rocprofv3 kernel stats of the bench command, measured HBM traffic, per-shape table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-c}
O=$PWD/gpurun_out; mkdir -p $O/${P}_benchprof
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider -s > $O/${P}_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/${P}_pytest.log)"
grep -h "\[unet \|headroom\|\[reduce+gn\|^FAILED" $O/${P}_pytest.log | sed 's/^\.*//' | cut -c1-170 | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; el "smoke exit $? : $(grep -h smoke: $O/${P}_smoke.log | head -3 | tr '\n' ' ')"
timeout 900 python bench.py > $O/${P}_bench.log 2>&1; el "bench exit $? : $(tail -1 $O/${P}_bench.log | cut -c1-200)"
tail -1 $O/${P}_bench.log > $O/${P}_bench_txt2img512.json
for w in txt2img768 img2img512; do
  timeout 600 python bench.py --workload $w --steps 4 --warmup 1 > $O/${P}_bench_$w.log 2>&1; el "bench $w exit $? : $(tail -1 $O/${P}_bench_$w.log | cut -c1-140)"
  tail -1 $O/${P}_bench_$w.log > $O/${P}_bench_$w.json
done
NCCL_DEBUG=INFO timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_torchrun_n1.log 2>&1; el "torchrun N=1 (RCCL) exit $? : $(tail -1 $O/${P}_torchrun_n1.log | cut -c1-120)"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${P}_benchprof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_prof.log 2>&1; el "rocprofv3 exit $?"
P=$P python - <<'PY'
import sqlite3, glob, os
P = os.environ['P']
for f in glob.glob(f'gpurun_out/{P}_benchprof/**/*_results.db', recursive=True):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open(f'gpurun_out/{P}_kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline   (2 images = 102 UNet calls + 2 first-stage decodes + the latency probes)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:70]:
            out.write(f'{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n')
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open(f'gpurun_out/{P}_kernel_stats.txt').read()[:1500])
PY
find $O/${P}_benchprof -type f ! -name '*.txt' ! -name '*.log' -delete; find $O -type d -empty -delete
timeout 600 python bench.py --traffic-pass --traffic-out $O/${P}_traffic.json > $O/${P}_traffic.log 2>&1; el "traffic pass exit $? : $(tail -1 $O/${P}_traffic.log | cut -c1-300)"
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes.txt 2>&1; el "prof_shapes exit $?"; grep -v amdgpu $O/${P}_shapes.txt | head -8
el done
