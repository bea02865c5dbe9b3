#!/bin/bash
# libsdmi_rctiming.so = the product objects + rowchain.hip / unet.cpp / api.cpp rebuilt with -DSDMI_RC_TIMING (cycle stamps and the
# timing-only ablations of ff_tail_kernel; FfTailParams grows a field, so its three users are recompiled).  For tools/ff_tail_timing.py.
set -e
cd "$(dirname "$0")/.."
python stable-diffusion_amd/build.py
S=stable-diffusion_amd/csrc; B=stable-diffusion_amd/build; T=stable-diffusion_amd/build_rctiming; mkdir -p $T
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-unused-variable -DSDMI_RC_TIMING"
/opt/rocm/bin/hipcc $F -x hip -c $S/rowchain.hip -o $T/rowchain.o &
/opt/rocm/bin/hipcc $F -x hip -c $S/gnconv.hip -o $T/gnconv.o &
/opt/rocm/bin/hipcc $F -c $S/unet.cpp -o $T/unet.o &
/opt/rocm/bin/hipcc $F -c $S/api.cpp -o $T/api.o &
wait
OBJS=$(ls $B/*.o | grep -v "/rowchain.o\|/gnconv.o\|/unet.o\|/api.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o stable-diffusion_amd/libsdmi_rctiming.so $OBJS $T/rowchain.o $T/gnconv.o $T/unet.o $T/api.o
cp stable-diffusion_amd/tune_gfx950.txt stable-diffusion_amd/tune_gfx950.txt 2>/dev/null || true
echo built stable-diffusion_amd/libsdmi_rctiming.so
