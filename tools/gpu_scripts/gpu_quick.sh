#!/bin/bash
# quick A/B: default bench + per-shape table with the committed tuning table (optionally a second table given as $1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O; P=${2:-q}
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_committed.log 2>&1; el "bench (committed table) exit $?"; tail -1 $O/${P}_bench_committed.log | cut -c1-200
timeout 600 python tools/prof_shapes.py > $O/${P}_shapes_committed.txt 2>&1; el "prof_shapes exit $?"; grep -E "^total|^attn|^splitk" $O/${P}_shapes_committed.txt
if [ -n "$1" ] && [ -f "$1" ]; then
SDMI_TUNE_FILE=$1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_alt.log 2>&1; el "bench ($1) exit $?"; tail -1 $O/${P}_bench_alt.log | cut -c1-200
fi
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_committed2.log 2>&1; el "bench (committed table, again) exit $?"; tail -1 $O/${P}_bench_committed2.log | cut -c1-200
el done
