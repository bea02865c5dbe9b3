#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) per kernel over UNet calls of the bench workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/pmc_unet; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/$c -o pmc -- python tools/prof_shapes.py > $O/$c.log 2>&1; echo "$c exit $?"
done
python - <<'PY'
import sqlite3, glob, collections
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
print(open('gpurun_out/pmc_unet/traffic_by_kernel.txt').read()[:3000])
PY
