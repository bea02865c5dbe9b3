#!/bin/bash
# Run on the GPU box (via gpurun): per-file GPU test logs + a short bench, all under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit:" | head -6 > gpurun_out/rocminfo.txt
nproc >> gpurun_out/rocminfo.txt; lscpu | grep "Model name" >> gpurun_out/rocminfo.txt
for f in ${TEST_FILES:-test_kernels_gpu test_unet_gpu test_sampler_gpu}; do
  timeout ${TEST_TIMEOUT:-1200} python -X faulthandler -m pytest tests/$f.py -m gpu -q -s -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/$f.log 2>&1
  echo "exit $?" >> gpurun_out/$f.log
  echo "== $f: $(grep -E 'passed|failed|error' gpurun_out/$f.log | tail -1) $(tail -1 gpurun_out/$f.log)"
done
if [ "${RUN_KBENCH:-0}" = "1" ]; then
  timeout 900 python tests/tools/bench_kernels.py ${KBENCH_ARGS} > gpurun_out/kbench.log 2>&1
  echo "kbench exit $?"; tail -45 gpurun_out/kbench.log
fi
if [ "${RUN_BENCH:-1}" = "1" ]; then
  timeout 900 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 ${BENCH_ARGS:---no-cpu-baseline} > gpurun_out/bench.log 2>&1
  echo "bench exit $?"; tail -3 gpurun_out/bench.log
fi
