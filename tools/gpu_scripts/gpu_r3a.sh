#!/bin/bash
# Round-3 GPU pass A (diagnostics): the whole GPU suite on the round-3 host-side changes, a baseline bench line on this box,
# per-workgroup phase stamps of every GEMM launch INCLUDING the halo-staged conv (timing build), epilogue ablations (timing
# build: no residual loads / no GroupNorm statistics / no output stores), the weight-prefetch experiment, torchrun N=1 with RCCL.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/a_pytest.log 2>&1; el "pytest -m gpu -x exit $? : $(tail -1 $O/a_pytest.log)"
grep -h "headroom\|TUNE_DISABLE\|ddim eta\|mask blend" $O/a_pytest.log | head
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/a_bench.log 2>&1; el "bench exit $?"; tail -1 $O/a_bench.log | cut -c1-400
TL=$PWD/stable-diffusion_amd/libsdmi_timing.so
SDMI_LIB_PATH=$TL timeout 600 python tools/igemm_timing.py $O/a_timing.txt > $O/a_timing.log 2>&1; el "phase stamps exit $?"; grep -c . $O/a_timing.txt
for abl in 0 1 2 4 3 7; do
  SDMI_LIB_PATH=$TL SDMI_EPI_ABL=$abl timeout 300 python tools/unet_latency.py "timing-lib EPI_ABL=$abl" 20 2 2>/dev/null | grep round >> $O/a_abl.txt
done
el "ablations done"; cat $O/a_abl.txt
timeout 600 python tools/bench_prefetch.py > $O/a_prefetch.txt 2>&1; el "prefetch exit $?"; cat $O/a_prefetch.txt | grep -v amdgpu.ids
NCCL_DEBUG=INFO timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/a_torchrun1.log 2>&1; el "torchrun N=1 (process group over RCCL, all_gather + all_reduce) exit $?"
grep -m 12 "NCCL INFO" $O/a_torchrun1.log | cut -c1-200; tail -1 $O/a_torchrun1.log | cut -c1-600
el done
