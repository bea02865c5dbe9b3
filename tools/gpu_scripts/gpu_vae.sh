#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_vae_gpu.py -m gpu -q -s -p no:cacheprovider > gpurun_out/test_vae_gpu.log 2>&1; echo "vae tests exit $?"; grep -E "passed|failed" gpurun_out/test_vae_gpu.log | tail -1
timeout 600 python tools/prof_vae.py 1 > gpurun_out/prof_vae.log 2>&1; echo "prof exit $?"; cat gpurun_out/prof_vae.log | grep -v amdgpu.ids
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_hipvae.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_hipvae.log | cut -c1-400
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --vae torch > gpurun_out/bench_torchvae.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_torchvae.log | cut -c1-400
