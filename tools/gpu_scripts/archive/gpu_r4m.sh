#!/bin/bash
# Round-4 GPU pass M: rocprofv3 PMC evidence of the final tree over UNet calls of the bench workload (tools/prof_shapes.py: model build + 2 UNet
# calls, CFG batch 2, 64x64): HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE, separate passes), SQ counters per kernel, kernel-trace durations
# of the same command, and from the three: MFMA utilisation and HBM GB/s per kernel.  Each counter set in its own run, no trace domains with --pmc.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/m_pmc; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/$c -o pmc -- python tools/prof_shapes.py > $O/$c.log 2>&1; el "$c exit $?"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/SQ -o pmc -- python tools/prof_shapes.py > $O/SQ.log 2>&1; el "SQ exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/KT -o kt -- python tools/prof_shapes.py > $O/KT.log 2>&1; el "kernel trace exit $?"
python - <<'PY'
import sqlite3, glob, collections
O='gpurun_out/m_pmc'
res=collections.defaultdict(dict)
for c in ('FETCH_SIZE','WRITE_SIZE','SQ'):
    for f in glob.glob(f'{O}/{c}/**/*_results.db', recursive=True):
        con=sqlite3.connect(f)
        for k,cn,v,n in con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            res[k][cn]=(v,n)
dur={}
for f in glob.glob(f'{O}/KT/**/*_results.db', recursive=True):
    con=sqlite3.connect(f)
    for name,calls,total,avg,pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        dur[name]=(calls,avg*1e3)      # top_kernels.average is in us -> ns
def short(k): return k.split('sdmi::(anonymous namespace)::')[-1].replace('sdmi::','')[:66]
with open('gpurun_out/m_pmc_by_kernel.txt','w') as out:
    out.write('rocprofv3 PMC passes over tools/prof_shapes.py (model build + 2 UNet calls, CFG batch 2, 64x64 latent), round-4 final tree, one MI355X box:\n')
    out.write('  --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES (separate runs),\n')
    out.write('  --kernel-trace --stats of the same command for the durations.  fetch_x2 = FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md);\n')
    out.write('  GB/s = (fetch_x2 + write) / average duration against 8000 GB/s; wait / stall / active = fractions of SQ_WAVE_CYCLES (parked on s_waitcnt or a barrier /\n')
    out.write('  issue stalls / issuing); mfma = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz): the share of the chip\'s MFMA issue time that was busy\n')
    out.write('  (an upper clock: the boxes run 1.9-2.3 GHz under load, so the true share is up to 1.25x higher).\n')
    out.write(f'{"kernel":68s} {"n":>5s} {"avg us":>8s} {"fetch_x2 MB":>11s} {"write MB":>9s} {"GB/s":>7s} {"of 8TB/s":>8s} {"wait":>6s} {"stall":>6s} {"active":>6s} {"mfma":>6s}\n')
    rows=[]
    for k,d in res.items():
        if 'sdmi' not in k or k not in dur or 'pack_' in k or 'ln_fold_prep' in k: continue
        calls,avg=dur[k]
        f,n=d.get('FETCH_SIZE',(0,1)); w,_=d.get('WRITE_SIZE',(0,1)); n=max(n,1)
        wc=max(d.get('SQ_WAVE_CYCLES',(1,1))[0],1); g=lambda c: d.get(c,(0,1))[0]
        fm, wm = 2*f/n/1024, w/n/1024
        gbs=(fm+wm)*1e6/ (avg*1e-9) /1e9 if avg else 0
        mf=g('SQ_VALU_MFMA_BUSY_CYCLES')/max(d.get('SQ_VALU_MFMA_BUSY_CYCLES',(0,1))[1],1)/(1024*avg*1e-9*2.4e9) if avg else 0
        rows.append((calls*avg, f'{short(k):68s} {calls:5d} {avg/1e3:8.2f} {fm:11.2f} {wm:9.2f} {gbs:7.0f} {gbs/8000:8.3f} {g("SQ_WAIT_ANY")/wc:6.3f} {g("SQ_WAIT_INST_ANY")/wc:6.3f} {g("SQ_ACTIVE_INST_ANY")/wc:6.3f} {mf:6.3f}'))
    for _,l in sorted(rows, reverse=True): out.write(l+'\n')
print(open('gpurun_out/m_pmc_by_kernel.txt').read()[:6000])
PY
find $O -type f ! -name '*.txt' ! -name '*.log' -delete; find $O -type d -empty -delete
el done
