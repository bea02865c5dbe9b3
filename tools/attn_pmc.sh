#!/bin/bash
# PMC passes over the attention kernel alone (d40, 4096 x 4096): pipelined, dma ring, register staged
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/attn_pmc; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
grep -oE "SQ_[A-Z_0-9]+" $O/avail.txt | sort -u > $O/sq_names.txt; wc -l $O/sq_names.txt
run() { # name, env, counters...
  local name=$1; shift; local envs=$1; shift
  env $envs timeout 300 rocprofv3 --pmc "$@" -d $O/$name -o pmc -- python tools/attn_one.py > $O/$name.log 2>&1; echo "$name exit $?"
}
for v in "pipe:SDMI_X=1" "dma:SDMI_ATTN_PIPE_MIN=1000000" "v1:SDMI_ATTN_V1=1"; do
  n=${v%%:*}; e=${v#*:}
  run ${n}_a $e SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
  run ${n}_b $e SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
  run ${n}_c $e SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run ${n}_d $e SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_MOPS_F16
done
python - <<'PY'
import glob, sqlite3, collections, os
O='gpurun_out/attn_pmc'
for d in sorted(glob.glob(O+'/*_[abcd]')):
    for f in glob.glob(d+'/**/*_results.db', recursive=True):
        try:
            con=sqlite3.connect(f)
            tabs=[r[0] for r in con.execute("select name from sqlite_master where type='table'")]
            pmc=[t for t in tabs if 'pmc_event' in t][0]; info=[t for t in tabs if 'info_pmc' in t][0]
            ksym=[t for t in tabs if 'kernel_symbol' in t][0]; disp=[t for t in tabs if 'kernel_dispatch' in t][0]
            q=f"""select s.kernel_name, i.name, sum(p.value), count(distinct p.event_id) from {pmc} p join {info} i on p.pmc_id=i.id
                 join {disp} k on p.event_id=k.event_id join {ksym} s on k.kernel_id=s.id group by s.kernel_name, i.name"""
            for kn, cn, v, n in con.execute(q):
                if 'attn' in kn: print(os.path.basename(d), kn[:60].replace('sdmi::(anonymous namespace)::',''), cn, '%.4g' % (v / max(n,1)), 'launches', n)
        except Exception as e:
            print('db', f, e)
PY
