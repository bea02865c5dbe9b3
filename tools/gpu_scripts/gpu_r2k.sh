#!/bin/bash
# kernel tests, re-tune in situ, same-box A/B of the new table vs the committed one, parity with the new table
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${1:-k}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider > $O/${P}_kernels.log 2>&1; el "kernel tests exit $? : $(tail -1 $O/${P}_kernels.log)"
grep -E "^FAILED|^ERROR" $O/${P}_kernels.log | head -30
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_committed.log 2>&1; el "bench (committed table) exit $?"; tail -1 $O/${P}_bench_committed.log | cut -c1-190
timeout 900 python tools/tune.py --reps ${TUNE_REPS:-3} --out $O/tune_gfx950.txt --dump $O/tune_dump.txt > $O/${P}_tune.log 2>&1; el "tune exit $? : $(tail -1 $O/${P}_tune.log)"
SDMI_TUNE_FILE=$O/tune_gfx950.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_new.log 2>&1; el "bench (new table) exit $?"; tail -1 $O/${P}_bench_new.log | cut -c1-190
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_committed2.log 2>&1; el "bench (committed table, again) exit $?"; tail -1 $O/${P}_bench_committed2.log | cut -c1-190
SDMI_TUNE_FILE=$O/tune_gfx950.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_new2.log 2>&1; el "bench (new table, again) exit $?"; tail -1 $O/${P}_bench_new2.log | cut -c1-190
export SDMI_TUNE_FILE=$O/tune_gfx950.txt
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_pipeline_gpu.py tests/test_vae_gpu.py tests/test_clip_gpu.py tests/test_sampler_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet/pipeline/vae/clip/sampler tests (new table) exit $? : $(tail -1 $O/${P}_unet.log)"
grep -E "^FAILED|^ERROR" $O/${P}_unet.log | head; grep -E "\[unet .*max-abs|\[pipeline|\[range" $O/${P}_unet.log | sed 's/^[.F]*//' | head -24
timeout 600 python tools/prof_shapes.py > $O/${P}_shapes.txt 2>&1; el "prof_shapes exit $?"; head -40 $O/${P}_shapes.txt
el done
