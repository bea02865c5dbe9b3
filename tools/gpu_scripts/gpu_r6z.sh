#!/bin/bash
# Round-6 closing evidence (final tree), ONE box: the driver's sequence (pytest -m gpu, smoke, bench with the driver's arguments), rocprofv3 --kernel-trace --stats
# of the bench command, then the PMC passes over UNet calls of the bench workload (tools/prof_shapes.py): FETCH_SIZE | WRITE_SIZE | SQ counters, each
# in its own run (no trace domains with --pmc), + a kernel trace of the same command for the durations -> per-kernel HBM GB/s and MFMA busy, and the
# traffic JSON bench.py reads (profiles/traffic_r06.json).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6z}
O=$PWD/gpurun_out; mkdir -p $O/${P}_benchprof $O/${P}_pmc
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/${P}_pytest.log 2>&1; el "pytest exit $? : $(tail -1 $O/${P}_pytest.log)"
timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet goldens (-s) exit $?"; grep -a "^.\?\[unet" $O/${P}_unet.log | cut -c1-260 > $O/${P}_unet_cases.txt; tail -3 $O/${P}_unet_cases.txt | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; el "smoke exit $?"; grep smoke: $O/${P}_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log > $O/${P}_bench.json; cut -c1-300 $O/${P}_bench.json
SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_exp.so timeout 600 python -m pytest tests -x -q -m "gpu and experiments" -p no:cacheprovider > $O/${P}_pytest_exp.log 2>&1; el "experiments library: pytest -m 'gpu and experiments' exit $? : $(tail -1 $O/${P}_pytest_exp.log)"
for w in txt2img768 img2img512; do
  timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/${P}_bench_$w.log 2>&1; el "bench $w exit $?"; tail -1 $O/${P}_bench_$w.log > $O/${P}_bench_$w.json; cut -c1-160 $O/${P}_bench_$w.json
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/${P}_benchprof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_benchprof.log 2>&1; el "bench kernel trace exit $?"
NCCL_DEBUG=INFO timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_torchrun.log 2>&1; el "torchrun N=1 (RCCL) exit $?"; grep -a "Init COMPLETE\|nranks" $O/${P}_torchrun.log | head -3; tail -1 $O/${P}_torchrun.log | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O/${P}_pmc/$c -o pmc -- python tools/prof_shapes.py > $O/${P}_pmc/$c.log 2>&1; el "$c exit $?"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/${P}_pmc/SQ -o pmc -- python tools/prof_shapes.py > $O/${P}_pmc/SQ.log 2>&1; el "SQ exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${P}_pmc/KT -o kt -- python tools/prof_shapes.py > $O/${P}_pmc/KT.log 2>&1; el "kernel trace exit $?"
grep "^total\|^splitk_reduce \|^groupnorm \|st_tail\|st_head\|st_mid\|attn_d40_self" $O/${P}_pmc/KT.log | cut -c1-140
python - "$P" <<'PY'
import sqlite3, glob, collections, json, sys
P = sys.argv[1]
O = f'gpurun_out/{P}_pmc'
# ---- kernel stats of the bench command
for f in glob.glob(f'gpurun_out/{P}_benchprof/**/*_results.db', recursive=True):
    con = sqlite3.connect(f)
    rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    with open(f'gpurun_out/{P}_kernel_stats.txt', 'w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline      (round-6 closing tree, the box of ' + P + '_bench.json)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name, calls, total, avg, pct in rows[:80]:
            out.write(f'{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n')
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
# ---- PMC per kernel
res = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ'):
    for f in glob.glob(f'{O}/{c}/**/*_results.db', recursive=True):
        con = sqlite3.connect(f)
        for k, cn, v, n in con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            res[k][cn] = (v, n)
dur = {}
for f in glob.glob(f'{O}/KT/**/*_results.db', recursive=True):
    con = sqlite3.connect(f)
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        dur[name] = (calls, avg * 1e3)
def short(k): return k.split('sdmi::(anonymous namespace)::')[-1].replace('sdmi::', '')[:70]
with open(f'gpurun_out/{P}_pmc_by_kernel.txt', 'w') as out:
    out.write('rocprofv3 PMC passes over tools/prof_shapes.py (model build + 2 UNet calls, CFG batch 2, 64x64 latent), round-6 closing tree, ONE MI355X box (the box of\n')
    out.write(f'  {P}_bench.json / kernel_stats_bench_r06.txt):  --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY\n')
    out.write('  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES (separate runs), --kernel-trace --stats of the same command for the durations.  fetch_x2 = FETCH_SIZE doubled (gfx950:\n')
    out.write('  128-byte requests tallied at 64 B, MI355X_MICROARCH.md); GB/s = (fetch_x2 + write) / average duration against 8000 GB/s; wait / stall / active = fractions of\n')
    out.write('  SQ_WAVE_CYCLES; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz) (an upper clock: the true share is up to 1.25x higher).\n')
    out.write(f'{"kernel":72s} {"n":>5s} {"avg us":>8s} {"fetch_x2 MB":>11s} {"write MB":>9s} {"GB/s":>7s} {"of 8TB/s":>8s} {"wait":>6s} {"stall":>6s} {"active":>6s} {"mfma":>6s}\n')
    rows = []
    for k, d in res.items():
        if 'sdmi' not in k or k not in dur or 'pack_' in k or 'ln_fold_prep' in k: continue
        calls, avg = dur[k]
        f, n = d.get('FETCH_SIZE', (0, 1)); w, _ = d.get('WRITE_SIZE', (0, 1)); n = max(n, 1)
        wc = max(d.get('SQ_WAVE_CYCLES', (1, 1))[0], 1); g = lambda c: d.get(c, (0, 1))[0]
        fm, wm = 2 * f / n / 1024, w / n / 1024
        gbs = (fm + wm) * 1e6 / (avg * 1e-9) / 1e9 if avg else 0
        mf = g('SQ_VALU_MFMA_BUSY_CYCLES') / max(d.get('SQ_VALU_MFMA_BUSY_CYCLES', (0, 1))[1], 1) / (1024 * avg * 1e-9 * 2.4e9) if avg else 0
        rows.append((calls * avg, f'{short(k):72s} {calls:5d} {avg/1e3:8.2f} {fm:11.2f} {wm:9.2f} {gbs:7.0f} {gbs/8000:8.3f} {g("SQ_WAIT_ANY")/wc:6.3f} {g("SQ_WAIT_INST_ANY")/wc:6.3f} {g("SQ_ACTIVE_INST_ANY")/wc:6.3f} {mf:6.3f}'))
    for _, l in sorted(rows, reverse=True): out.write(l + '\n')
# ---- traffic per kernel family (what bench.py --traffic-pass writes)
def family(name):
    for key, fam in (('igemm_kernel', 'igemm_family'), ('igemm5_kernel', 'igemm_family'), ('conv3halo_kernel', 'igemm_family'), ('conv3halo_gn_kernel', 'igemm_family'),
                     ('gemm_split16_kernel', 'igemm_family'), ('ff_tail_kernel', 'igemm_family'), ('st_head_kernel', 'igemm_family'), ('gn_conv3_kernel', 'igemm_family'),
                     ('attn', 'attention'), ('splitk_reduce', 'splitk_reduce'), ('gn_apply', 'groupnorm'), ('gn_stats', 'groupnorm'), ('layernorm', 'layernorm')):
        if key in name: return fam
    return None
kern = {}
for k, d in res.items():
    fam = family(k)
    if fam is None or 'FETCH_SIZE' not in d: continue
    e = kern.setdefault(fam, {'launches': 0, 'fetch_kb_x2': 0.0, 'write_kb': 0.0})
    e['launches'] += d['FETCH_SIZE'][1]; e['fetch_kb_x2'] += 2.0 * d['FETCH_SIZE'][0]; e['write_kb'] += d.get('WRITE_SIZE', (0, 1))[0]
for e in kern.values():
    e['fetch_mb_x2'] = e.pop('fetch_kb_x2') / 1024.0 / e['launches']; e['write_mb'] = e.pop('write_kb') / 1024.0 / e['launches']
json.dump({'source': 'tools/gpu_scripts/gpu_r6z.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/prof_shapes.py (model build + UNet calls, CFG batch 2, 64x64), the box of the closing bench line',
           'kernels': kern}, open(f'gpurun_out/{P}_traffic.json', 'w'), indent=1, sort_keys=True)
print(open(f'gpurun_out/{P}_pmc_by_kernel.txt').read()[:4500])
print(json.dumps(kern)[:600])
PY
cp $O/${P}_pmc/KT.log $O/${P}_shapes.txt 2>/dev/null
rm -rf $O/${P}_benchprof $O/${P}_pmc
el done
