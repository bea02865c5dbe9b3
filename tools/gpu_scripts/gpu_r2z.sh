#!/bin/bash
# Round-2 final evidence pass: the driver's own sequence (pytest -m gpu -x, smoke, default bench) + bench lines for the 768 /
# img2img workloads + torchrun N=1 + rocprofv3 kernel stats + PMC passes (HBM traffic via bench.py --traffic-pass, SQ waits /
# MFMA busy incl. per-head-dim attention utilisation).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O/z_prof $O/z_pmc
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/z_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/z_pytest.log)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/z_smoke.log 2>&1; el "smoke exit $?"; grep smoke: $O/z_smoke.log
timeout 600 python bench.py --traffic-pass --traffic-out $O/z_traffic.json > $O/z_traffic.log 2>&1; el "traffic pass exit $?"; tail -1 $O/z_traffic.log | cut -c1-300
cp $O/z_traffic.json profiles/traffic_r02.json 2>/dev/null
timeout 900 python bench.py > $O/z_bench.log 2>&1; el "bench exit $?"; tail -1 $O/z_bench.log | cut -c1-300
timeout 900 python bench.py --workload txt2img768 --steps 2 --warmup 1 > $O/z_bench768.log 2>&1; el "bench 768 exit $?"; tail -1 $O/z_bench768.log | cut -c1-200
timeout 900 python bench.py --workload img2img512 --steps 3 --warmup 1 > $O/z_benchi2i.log 2>&1; el "bench img2img exit $?"; tail -1 $O/z_benchi2i.log | cut -c1-200
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/z_torchrun1.log 2>&1; el "torchrun N=1 (RCCL init + all_gather path) exit $?"; tail -1 $O/z_torchrun1.log | cut -c1-200
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
timeout 900 rocprofv3 --kernel-trace --stats -d $O/z_prof/b -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/z_prof_b.log 2>&1; el "rocprof bench exit $?"
stats $O/z_prof/b $O/z_kernel_stats_bench.txt "rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline   (2 images = 102 UNet calls + 2 first-stage decodes + 12 UNet calls / 6 decodes of the latency probes)"
head -24 $O/z_kernel_stats_bench.txt | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats -d $O/z_prof/c -o bench -- python bench.py --workload txt2img768 --steps 1 --warmup 1 --no-roofline > $O/z_prof_768.log 2>&1; el "rocprof 768 exit $?"
stats $O/z_prof/c $O/z_kernel_stats_768.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload txt2img768 --steps 1 --warmup 1 --no-roofline   (2 images = 100 UNet calls at latent 96x96 + 2 decodes + probes)"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/z_prof/d -o bench -- python bench.py --workload img2img512 --steps 1 --warmup 1 --no-roofline > $O/z_prof_i2i.log 2>&1; el "rocprof img2img exit $?"
stats $O/z_prof/d $O/z_kernel_stats_img2img.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload img2img512 --steps 1 --warmup 1 --no-roofline   (2 images = 2 encodes + 74 UNet calls + 2 decodes + probes)"
timeout 600 python tools/prof_shapes.py > $O/z_shapes.txt 2>&1; el "prof_shapes exit $?"; head -3 $O/z_shapes.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/z_pmc/SQ -o pmc -- python tools/prof_shapes.py > $O/z_pmc/SQ.log 2>&1; el "SQ exit $?"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM -d $O/z_pmc/SQ2 -o pmc -- python tools/prof_shapes.py > $O/z_pmc/SQ2.log 2>&1; el "SQ2 exit $?"
python - <<'PY'
import sqlite3, glob, collections, json, os, re
def load(sub):
    res = collections.defaultdict(dict)
    for f in glob.glob(f'gpurun_out/z_pmc/{sub}/**/*_results.db', recursive=True):
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
sq = load('SQ')
# durations per attention head dim from the HIP-event table of the same script (one UNet call; the PMC run made 2)
dur = {}
for l in open('gpurun_out/z_shapes.txt'):
    m = re.match(r'attn_d(\d+)_(self|ctx)\s+n=\s*(\d+) total\s+([\d.]+) us', l)
    if m: dur[int(m.group(1))] = dur.get(int(m.group(1)), 0.0) + float(m.group(4))
with open('gpurun_out/z_pmc_sq_by_kernel.txt', 'w') as out:
    out.write('rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES over tools/prof_shapes.py\n')
    out.write('(2 UNet calls, CFG batch 2, 64x64).  WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stalls, ACTIVE = issuing; fractions of SQ_WAVE_CYCLES;\n')
    out.write('mfma_Mcyc = SQ_VALU_MFMA_BUSY_CYCLES per launch / 1e6 (summed over the SIMDs; utilisation = that / (1024 SIMDs x duration x clock))\n')
    out.write(f'{"kernel":66s} {"launches":>8s} {"wait_any":>9s} {"wait_inst":>9s} {"active":>8s} {"mfma_Mcyc":>9s}\n')
    att = collections.defaultdict(float)
    for k, d in sorted(sq.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', (0, 1))[0]):
        if 'sdmi' not in k: continue
        wc, n = d.get('SQ_WAVE_CYCLES', (1, 1))
        g = lambda c: d.get(c, (0, 1))[0]
        out.write(f'{short(k):66s} {n:8d} {g("SQ_WAIT_ANY")/wc:9.3f} {g("SQ_WAIT_INST_ANY")/wc:9.3f} {g("SQ_ACTIVE_INST_ANY")/wc:8.3f} {g("SQ_VALU_MFMA_BUSY_CYCLES")/n/1e6:9.3f}\n')
        m = re.search(r'attn_\w*kernel<(\d+),', k)
        if m: att[int(m.group(1))] += g('SQ_VALU_MFMA_BUSY_CYCLES') / 2.0       # per UNet call
    out.write('\nattention, MFMA pipe utilisation per head dim (all launches of a UNet call: self + cross attention):\n')
    out.write('  busy = SQ_VALU_MFMA_BUSY_CYCLES per call; time = HIP-event durations of the same launches (gpurun_out/z_shapes.txt); 1024 SIMDs, 2.4 GHz\n')
    for dh in sorted(att):
        if dh in dur:
            out.write(f'  d = {dh:3d}: busy {att[dh]/1e6:8.2f} Mcycles, {dur[dh]:7.1f} us  ->  {att[dh] / (1024 * dur[dh] * 1e-6 * 2.4e9):.3f} of the MFMA issue slots\n')
print(open('gpurun_out/z_pmc_sq_by_kernel.txt').read()[:3600])
sq2 = load('SQ2')
with open('gpurun_out/z_pmc_lds_by_kernel.txt', 'w') as out:
    out.write('rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM over tools/prof_shapes.py (per launch)\n')
    out.write(f'{"kernel":66s} {"launches":>8s} {"bank_conf":>10s} {"lds_active":>10s} {"valu":>10s} {"mfma":>10s} {"lds":>10s} {"vmem":>10s}\n')
    for k, d in sorted(sq2.items(), key=lambda kv: -kv[1].get('SQ_INSTS_MFMA', (0, 1))[0]):
        if 'sdmi' not in k: continue
        n = max(v[1] for v in d.values())
        g = lambda c: d.get(c, (0, 1))[0] / n
        out.write(f'{short(k):66s} {n:8d} {g("SQ_LDS_BANK_CONFLICT"):10.0f} {g("SQ_LDS_IDX_ACTIVE"):10.0f} {g("SQ_INSTS_VALU"):10.0f} {g("SQ_INSTS_MFMA"):10.0f} {g("SQ_INSTS_LDS"):10.0f} {g("SQ_INSTS_VMEM"):10.0f}\n')
print(open('gpurun_out/z_pmc_lds_by_kernel.txt').read()[:1500])
PY
el done
