#!/bin/bash
# PMC / kernel-trace passes over one igemm shape (each counter set in its own run; no trace domains with --pmc)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/prof; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
CMD="python tools/prof_igemm.py --shape ${SHAPE:-l0conv} --tiles ${TILES:-3,5} --iters 10"
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $CMD > $O/trace.log 2>&1
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/pmc$i -o pmc -- $CMD > $O/pmc$i.log 2>&1
  echo "pmc$i ($set) exit $?"
done <<SETS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES
SETS
find $O -name "*.csv" | head -40
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get('O','gpurun_out/prof')
for f in sorted(glob.glob('gpurun_out/prof/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','')[:60]
        if 'igemm' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    print('==',f)
    for k,v in agg.items():
        print(k, {c: round(x/10) for c,x in v.items()})
for f in sorted(glob.glob('gpurun_out/prof/**/*kernel_stats.csv', recursive=True)):
    print('==',f); print(open(f).read()[:1500])
PY
