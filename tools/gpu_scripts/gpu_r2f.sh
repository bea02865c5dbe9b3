#!/bin/bash
# Round-2 evidence pass: the driver's own sequence (pytest -m gpu -x, smoke, default bench) + bench lines for the 768 /
# img2img workloads + torchrun N=1 + rocprofv3 kernel stats + PMC passes (HBM traffic, SQ waits / MFMA busy).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O/f_prof $O/f_pmc
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/f_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/f_pytest.log)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/f_smoke.log 2>&1; el "smoke exit $?"; grep smoke: $O/f_smoke.log
fi
timeout 900 python bench.py > $O/f_bench.log 2>&1; el "bench exit $?"; tail -1 $O/f_bench.log | cut -c1-300
timeout 900 python bench.py --workload txt2img768 --steps 2 --warmup 1 > $O/f_bench768.log 2>&1; el "bench 768 exit $?"; tail -1 $O/f_bench768.log | cut -c1-200
timeout 900 python bench.py --workload img2img512 --steps 3 --warmup 1 > $O/f_benchi2i.log 2>&1; el "bench img2img exit $?"; tail -1 $O/f_benchi2i.log | cut -c1-200
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/f_torchrun1.log 2>&1; el "torchrun N=1 (RCCL init + all_gather path) exit $?"; tail -1 $O/f_torchrun1.log | cut -c1-200
stats() {  # $1 = db dir, $2 = out file, $3 = header
python - "$1" "$2" "$3" <<'PY'
import sqlite3, glob, sys, os
d, outp, hdr = sys.argv[1:4]
for f in glob.glob(d + '/**/*_results.db', recursive=True):
    con = sqlite3.connect(f)
    rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    with open(outp, 'w') as out:
        out.write(hdr + '\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name, calls, total, avg, pct in rows[:70]:
            out.write(f"{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n")
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    os.remove(f)
PY
}
timeout 900 rocprofv3 --kernel-trace --stats -d $O/f_prof/b -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/f_prof_b.log 2>&1; el "rocprof bench exit $?"
stats $O/f_prof/b $O/f_kernel_stats_bench.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline   (2 images = 102 UNet calls + 2 first-stage decodes + 12 UNet calls / 6 decodes of the latency probes)"
head -30 $O/f_kernel_stats_bench.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $O/f_prof/c -o bench -- python bench.py --workload txt2img768 --steps 1 --warmup 1 --no-roofline > $O/f_prof_768.log 2>&1; el "rocprof 768 exit $?"
stats $O/f_prof/c $O/f_kernel_stats_768.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload txt2img768 --steps 1 --warmup 1 --no-roofline   (2 images = 100 UNet calls at latent 96x96 + 2 decodes + probes)"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/f_prof/d -o bench -- python bench.py --workload img2img512 --steps 1 --warmup 1 --no-roofline > $O/f_prof_i2i.log 2>&1; el "rocprof img2img exit $?"
stats $O/f_prof/d $O/f_kernel_stats_img2img.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload img2img512 --steps 1 --warmup 1 --no-roofline   (2 images = 2 encodes + 74 UNet calls + 2 decodes + probes)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/f_pmc/$c -o pmc -- python tools/prof_shapes.py > $O/f_pmc/$c.log 2>&1; el "$c exit $?"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/f_pmc/SQ -o pmc -- python tools/prof_shapes.py > $O/f_pmc/SQ.log 2>&1; el "SQ exit $?"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM -d $O/f_pmc/SQ2 -o pmc -- python tools/prof_shapes.py > $O/f_pmc/SQ2.log 2>&1; el "SQ2 exit $?"
python - <<'PY'
import sqlite3, glob, collections, json, os
def load(sub):
    res = collections.defaultdict(dict)
    for f in glob.glob(f'gpurun_out/f_pmc/{sub}/**/*_results.db', recursive=True):
        con = sqlite3.connect(f)
        try:
            q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"
            for k, c, v, n in con.execute(q):
                res[k][c] = (v, n)
        except Exception as e:
            print('pmc db', f, e)
        os.remove(f)
    return res
def short(k):
    k = k.split('sdmi::(anonymous namespace)::')[-1]
    return k.replace('(sdmi::IGemmParams, int, int, int)', '').replace('void ', '')[:66]
res = collections.defaultdict(dict)
for sub in ('FETCH_SIZE', 'WRITE_SIZE'):
    for k, d in load(sub).items():
        res[k].update(d)
traffic = {}
with open('gpurun_out/f_hbm_traffic_by_kernel.txt', 'w') as out:
    out.write('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/prof_shapes.py (model build + 2 UNet calls, CFG batch 2, 64x64)\n')
    out.write('values in KiB as reported; gfx950 note (MI355X_MICROARCH.md HBM): FETCH_SIZE under-reports wide coalesced reads by 2x -> "fetch_x2" column\n')
    out.write(f'{"kernel":68s} {"launches":>8s} {"fetch MB/launch":>16s} {"fetch_x2":>10s} {"write MB/launch":>16s}\n')
    for k, d in sorted(res.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', (0, 1))[0]):
        if 'sdmi' not in k: continue
        f, n = d.get('FETCH_SIZE', (0, 1)); w, _ = d.get('WRITE_SIZE', (0, 1))
        out.write(f'{short(k):68s} {n:8d} {f/n/1024:16.2f} {2*f/n/1024:10.2f} {w/n/1024:16.2f}\n')
        traffic[short(k)] = {'launches': n, 'fetch_mb_x2': 2 * f / n / 1024, 'write_mb': w / n / 1024}
fam = [v for k, v in traffic.items() if k.startswith(('igemm_kernel', 'conv3halo_kernel'))]
if fam:
    nl = sum(v['launches'] for v in fam)
    traffic['igemm_family'] = {'launches': nl, 'fetch_mb_x2': sum(v['fetch_mb_x2'] * v['launches'] for v in fam) / nl,
                               'write_mb': sum(v['write_mb'] * v['launches'] for v in fam) / nl}
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/prof_shapes.py (round 2, tools/gpu_r2f.sh)', 'kernels': traffic},
          open('gpurun_out/f_traffic.json', 'w'), indent=1)
print(open('gpurun_out/f_hbm_traffic_by_kernel.txt').read()[:3000])
sq = load('SQ')
with open('gpurun_out/f_pmc_sq_by_kernel.txt', 'w') as out:
    out.write('rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES over tools/prof_shapes.py\n')
    out.write('(2 UNet calls, CFG batch 2, 64x64).  WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stalls, ACTIVE = issuing; fractions of SQ_WAVE_CYCLES;\n')
    out.write('mfma_Mcyc = SQ_VALU_MFMA_BUSY_CYCLES per launch / 1e6 (summed over the SIMDs; utilisation = that / (1024 SIMDs x duration x clock))\n')
    out.write(f'{"kernel":66s} {"launches":>8s} {"wait_any":>9s} {"wait_inst":>9s} {"active":>8s} {"mfma_Mcyc":>9s}\n')
    for k, d in sorted(sq.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', (0, 1))[0]):
        if 'sdmi' not in k: continue
        wc, n = d.get('SQ_WAVE_CYCLES', (1, 1))
        g = lambda c: d.get(c, (0, 1))[0]
        out.write(f'{short(k):66s} {n:8d} {g("SQ_WAIT_ANY")/wc:9.3f} {g("SQ_WAIT_INST_ANY")/wc:9.3f} {g("SQ_ACTIVE_INST_ANY")/wc:8.3f} {g("SQ_VALU_MFMA_BUSY_CYCLES")/n/1e6:9.3f}\n')
print(open('gpurun_out/f_pmc_sq_by_kernel.txt').read()[:3000])
sq2 = load('SQ2')
with open('gpurun_out/f_pmc_lds_by_kernel.txt', 'w') as out:
    out.write('rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM over tools/prof_shapes.py (per launch)\n')
    out.write(f'{"kernel":66s} {"launches":>8s} {"bank_conf":>10s} {"lds_active":>10s} {"valu":>10s} {"mfma":>10s} {"lds":>10s} {"vmem":>10s}\n')
    for k, d in sorted(sq2.items(), key=lambda kv: -kv[1].get('SQ_INSTS_MFMA', (0, 1))[0]):
        if 'sdmi' not in k: continue
        n = max(v[1] for v in d.values())
        g = lambda c: d.get(c, (0, 1))[0] / n
        out.write(f'{short(k):66s} {n:8d} {g("SQ_LDS_BANK_CONFLICT"):10.0f} {g("SQ_LDS_IDX_ACTIVE"):10.0f} {g("SQ_INSTS_VALU"):10.0f} {g("SQ_INSTS_MFMA"):10.0f} {g("SQ_INSTS_LDS"):10.0f} {g("SQ_INSTS_VMEM"):10.0f}\n')
print(open('gpurun_out/f_pmc_lds_by_kernel.txt').read()[:2500])
PY
el done
