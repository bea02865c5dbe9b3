#!/bin/bash
# A second libsdmi next to the product one that differs in ONE translation unit (bisecting / instrumented builds without
# recompiling everything): tools/build_variant.sh <lib name> <source in csrc> <extra flags...>
#   tools/build_variant.sh libsdmi_gnvis.so conv3halo.hip -DSDMI_GN_VISIBLE     -> SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_gnvis.so
# The other objects come from stable-diffusion_amd/build/ (run stable-diffusion_amd/build.py first).
set -e
cd "$(dirname "$0")/.."
LIB=$1; SRC=$2; shift 2
P=stable-diffusion_amd
O=$P/build_variant_${LIB%.so}; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed"
/opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c $P/csrc/$SRC -o $O/${SRC%.*}.o
OBJS=""
for f in $P/build/*.o; do
  b=$(basename $f)
  if [ "$b" = "${SRC%.*}.o" ]; then OBJS="$OBJS $O/$b"; else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/$LIB $OBJS
echo "built $P/$LIB"
