#!/bin/bash
# Round-4 GPU pass Q (closing): the driver's own sequence on the committed tree -- pytest -m gpu -x, smoke, bench.py with the driver's arguments
# (--gpus 1 --steps 20 --warmup 5) -- and the UNet latency per CFG batch (tools/bench_batch.py).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-q}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/${P}_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/${P}_pytest.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; el "smoke exit $? : $(grep -h smoke: $O/${P}_smoke.log | head -3 | tr '\n' ' ')"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${P}_bench.log 2>&1; el "bench (driver arguments) exit $? : $(tail -1 $O/${P}_bench.log | cut -c1-260)"
tail -1 $O/${P}_bench.log > $O/${P}_bench_driver_cmd.json
timeout 600 python tools/bench_batch.py > $O/${P}_batch.txt 2>&1; el "bench_batch exit $?"; grep -v amdgpu $O/${P}_batch.txt
el done
