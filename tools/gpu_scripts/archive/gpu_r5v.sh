#!/bin/bash
# Round-5 GPU pass V: the whole GPU suite + smoke + the driver's bench command on the tree with the igemm instantiations in three files
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5v}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=15 > $O/${P}_pytest.log 2>&1; el "pytest exit $? : $(tail -1 $O/${P}_pytest.log)"
grep -A18 "slowest" $O/${P}_pytest.log | cut -c1-150
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${P}_smoke.log 2>&1; el "smoke exit $?"; grep smoke: $O/${P}_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log > $O/${P}_bench.json; cut -c1-700 $O/${P}_bench.json
el done
