#!/bin/bash
# Round 6, pass K: the ResBlock skip convolutions on a side stream (round 2: measured slower) re-measured on the round-6 kernels (experiments library,
# SDMI_SIDE_STREAM=1; the launch tapes are off on that path).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
X=$PWD/stable-diffusion_amd/libsdmi_exp.so
for rep in 1 2 3; do
  SDMI_LIB_PATH=$X timeout 300 python tools/unet_latency.py "experiments lib, one stream" 20 2 2>&1 | grep round
  SDMI_LIB_PATH=$X SDMI_SIDE_STREAM=1 timeout 300 python tools/unet_latency.py "skip convs on a side stream" 20 2 2>&1 | grep round
done | tee gpurun_out/r6k_side.log
