#!/bin/bash
# Round-end verification on one box: the driver's own sequence (pytest -m gpu -x, smoke, default bench) + rocprofv3 summaries.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O/benchprof $O/pmc_unet
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit $? : $(tail -1 $O/pytest_gpu.log)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; grep smoke: $O/smoke.log
fi
timeout 900 python bench.py > $O/bench_full.log 2>&1; echo "bench exit $?"; tail -1 $O/bench_full.log | cut -c1-300
timeout 900 rocprofv3 --kernel-trace --stats -d $O/benchprof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/benchprof.log 2>&1; echo "prof exit $?"
python - <<'PY'
import sqlite3, glob
for f in glob.glob('gpurun_out/benchprof/*_results.db'):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open('gpurun_out/benchprof/kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline\n(2 images = 102 UNet calls + 2 first-stage decodes + 21 UNet calls / 6 decodes of the latency probes)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:60]:
            out.write(f'{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n')
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open('gpurun_out/benchprof/kernel_stats.txt').read()[:2500])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/pmc_unet/$c -o pmc -- python tools/prof_shapes.py > $O/pmc_unet/$c.log 2>&1; echo "$c exit $?"
done
python - <<'PY'
import sqlite3, glob, collections, json
res=collections.defaultdict(dict)
for c in ('FETCH_SIZE','WRITE_SIZE'):
    for f in glob.glob(f'gpurun_out/pmc_unet/{c}/*_results.db'):
        con=sqlite3.connect(f)
        for k,v,n in con.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name=? group by kernel_name",(c,)):
            res[k][c]=(v,n)
with open('gpurun_out/pmc_unet/traffic_by_kernel.txt','w') as out:
    out.write('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/prof_shapes.py (model build + 2 UNet calls, CFG batch 2, 64x64)\n')
    out.write('values in KiB as reported; gfx950 note (MI355X_MICROARCH.md HBM): FETCH_SIZE under-reports wide coalesced reads by 2x -> "fetch_x2" column\n')
    out.write(f'{"kernel":70s} {"launches":>8s} {"fetch MB/launch":>16s} {"fetch_x2":>10s} {"write MB/launch":>16s}\n')
    for k,d in sorted(res.items(), key=lambda kv:-kv[1].get('FETCH_SIZE',(0,1))[0]):
        if 'sdmi' not in k: continue
        f,n=d.get('FETCH_SIZE',(0,1)); w,_=d.get('WRITE_SIZE',(0,1))
        name=k.split('sdmi::(anonymous namespace)::')[-1][:68]
        out.write(f'{name:70s} {n:8d} {f/n/1024:16.2f} {2*f/n/1024:10.2f} {w/n/1024:16.2f}\n')
print(open('gpurun_out/pmc_unet/traffic_by_kernel.txt').read()[:2500])
PY
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/pmc_unet/SQ -o pmc -- python tools/prof_shapes.py > $O/pmc_unet/SQ.log 2>&1; echo "SQ exit $?"
python - <<'PY'
import sqlite3, glob, collections
res=collections.defaultdict(dict)
for f in glob.glob('gpurun_out/pmc_unet/SQ/*_results.db'):
    con=sqlite3.connect(f)
    for k,c,v,n in con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        res[k][c]=(v,n)
with open('gpurun_out/pmc_unet/sq_by_kernel.txt','w') as out:
    out.write('rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES over tools/prof_shapes.py\n')
    out.write('(2 UNet calls, CFG batch 2, 64x64).  WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stalls, ACTIVE = issuing; fractions of SQ_WAVE_CYCLES;\n')
    out.write('mfma_Mcyc = SQ_VALU_MFMA_BUSY_CYCLES per launch / 1e6 (summed over the SIMDs; utilisation = that / (1024 SIMDs x duration x clock))\n')
    out.write(f'{"kernel":64s} {"launches":>8s} {"wait_any":>9s} {"wait_inst":>9s} {"active":>8s} {"mfma_Mcyc":>9s}\n')
    for k,d in sorted(res.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',(0,1))[0]):
        if 'sdmi' not in k: continue
        wc,n=d.get('SQ_WAVE_CYCLES',(1,1)); wc=max(wc,1)
        g=lambda c: d.get(c,(0,1))[0]
        name=k.split('sdmi::(anonymous namespace)::')[-1][:62]
        out.write(f'{name:64s} {n:8d} {g("SQ_WAIT_ANY")/wc:9.3f} {g("SQ_WAIT_INST_ANY")/wc:9.3f} {g("SQ_ACTIVE_INST_ANY")/wc:8.3f} {g("SQ_VALU_MFMA_BUSY_CYCLES")/n/1e6:9.3f}\n')
print(open('gpurun_out/pmc_unet/sq_by_kernel.txt').read()[:2500])
PY
# keep only the text summaries (the rocprofv3 databases exceed the 64 MiB gpurun_out budget)
find $O/benchprof $O/pmc_unet -type f ! -name '*.txt' ! -name '*.log' -delete; find $O -type d -empty -delete; du -sh $O
