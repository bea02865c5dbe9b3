#!/bin/bash
# Round-3 closing check on one box, the driver's own sequence on the committed tree: pytest -m gpu -x, smoke, default bench.py.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/p_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/p_pytest.log)"
grep -h "^FAILED" $O/p_pytest.log | head
timeout 600 python __graft_entry__.py --smoke > $O/p_smoke.log 2>&1; el "smoke exit $? : $(grep -h smoke $O/p_smoke.log | head -3 | tr '\n' '|' | cut -c1-300)"
timeout 900 python bench.py > $O/p_bench.log 2>&1; el "bench exit $? : $(tail -1 $O/p_bench.log | cut -c1-200)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/p_torchrun1.log 2>&1; el "torchrun N=1 exit $? : $(tail -1 $O/p_torchrun1.log | cut -c60-130)"
python - <<'PY'
import os, glob
# which shared objects did the test process map?  (the driver records the same)
print('libsdmi.so present:', os.path.exists('stable-diffusion_amd/libsdmi.so'), os.path.getsize('stable-diffusion_amd/libsdmi.so'))
PY
el done
