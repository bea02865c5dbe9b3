#!/bin/bash
# Round-2 GPU pass B: kernel tests of every tile (+ GroupNorm statistics in the epilogue), the whole GPU suite with the
# committed tuning table, GN-fusion A/B, accurate kernel durations (rocprofv3), the 768 / img2img workloads, torchrun N=1.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "igemm" -p no:cacheprovider > $O/b_igemm.log 2>&1; el "igemm tests exit $? : $(tail -1 $O/b_igemm.log)"
grep -E "^FAILED|^ERROR" $O/b_igemm.log | head -30
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_kernels_gpu.py > $O/b_pytest.log 2>&1; el "gpu suite (rest) exit $? : $(tail -1 $O/b_pytest.log)"
grep -E "^FAILED|^ERROR" $O/b_pytest.log | head -20
grep -E "^\[unet |^\[pipeline" $O/b_pytest.log | head -30
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "not igemm" -p no:cacheprovider > $O/b_kernels_rest.log 2>&1; el "other kernel tests exit $? : $(tail -1 $O/b_kernels_rest.log)"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b_bench.log 2>&1; el "bench exit $?"; tail -1 $O/b_bench.log | cut -c1-400
SDMI_FUSE_GN_STATS=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/b_bench_nofuse.log 2>&1; el "bench (GN stats kernels) exit $?"; tail -1 $O/b_bench_nofuse.log | cut -c1-300
timeout 600 python tools/prof_shapes.py > $O/b_shapes.txt 2>&1; el "prof_shapes exit $?"; head -12 $O/b_shapes.txt
timeout 900 python bench.py --workload txt2img768 --steps 2 --warmup 1 > $O/b_bench768.log 2>&1; el "bench 768 exit $?"; tail -1 $O/b_bench768.log | cut -c1-400
timeout 900 python bench.py --workload img2img512 --steps 2 --warmup 1 > $O/b_benchi2i.log 2>&1; el "bench img2img exit $?"; tail -1 $O/b_benchi2i.log | cut -c1-400
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/b_torchrun1.log 2>&1; el "torchrun N=1 exit $?"; tail -1 $O/b_torchrun1.log | cut -c1-300
mkdir -p $O/b_prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/b_prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/b_prof.log 2>&1; el "rocprof exit $?"
python - <<'PY'
import sqlite3, glob
for f in glob.glob('gpurun_out/b_prof/**/*_results.db', recursive=True):
    con=sqlite3.connect(f)
    rows=con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    tot=sum(r[2] for r in rows)
    with open('gpurun_out/b_kernel_stats.txt','w') as out:
        out.write('rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline\n(2 images = 102 UNet calls + 2 first-stage decodes + 12 UNet calls / 6 decodes of the latency probes)\n')
        out.write(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"pct":>6s}  kernel\n')
        for name,calls,total,avg,pct in rows[:70]:
            out.write(f"{calls:7d} {total/1e3:10.3f} {avg:9.2f} {pct:6.2f}  {name[:150]}\n")
        out.write(f'total kernel time {tot/1e3:.1f} ms\n')
    print(open('gpurun_out/b_kernel_stats.txt').read()[:3500])
PY
el done
