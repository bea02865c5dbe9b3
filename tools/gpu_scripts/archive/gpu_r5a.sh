#!/bin/bash
# Round-5 GPU pass A: first contact of the row-strip chain kernel (ff_tail): its parity tests, the UNet goldens with it on, the
# microbenchmark against the three launches, the UNet latency A/B (SDMI_FF_TAIL=0 / 1, interleaved) and a short bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5a}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_rowchain_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_rowchain.log 2>&1; el "rowchain tests exit $? : $(tail -1 $O/${P}_rowchain.log)"
grep -h "ff_tail" $O/${P}_rowchain.log | head -30
timeout 300 python tools/bench_ff_tail.py 50 > $O/${P}_ff_tail_bench.txt 2>&1; el "bench_ff_tail exit $?"; grep -v amdgpu $O/${P}_ff_tail_bench.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet goldens exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "headroom\|max-abs" $O/${P}_unet.log | head -30
timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -p no:cacheprovider -s > $O/${P}_pipeline.log 2>&1; el "pipeline exit $? : $(tail -1 $O/${P}_pipeline.log)"
grep -h "\[pipeline" $O/${P}_pipeline.log
for r in 1 2; do
  SDMI_FF_TAIL=0 timeout 300 python tools/unet_latency.py "three launches" 20 2 2>&1 | grep -v amdgpu
  SDMI_FF_TAIL=1 timeout 300 python tools/unet_latency.py "ff_tail chain" 20 2 2>&1 | grep -v amdgpu
done
timeout 600 python bench.py --steps 3 --warmup 1 > $O/${P}_bench.log 2>&1; el "bench exit $? : $(tail -1 $O/${P}_bench.log | cut -c1-400)"
el done
