#!/bin/bash
# Round-4 GPU pass A: (1) GroupNorm + SiLU applied by the split-K reduction (splitk_reduce_gn_kernel, SDMI_REDUCE_GN): kernel tests,
# UNet goldens, same-box A/B against the two launches; (2) the unmodified reference scripts (bytecode bundle) driving the HIP path;
# (3) write-through output stores (-DSDMI_WT_STORES=1|2 builds: sc1 on the 16-byte fp32 / also the 8-byte fp16 stores) and the
# GroupNorm-apply 4-quads-per-thread threshold as same-box A/Bs; (4) the whole GPU suite, bench, per-shape table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-a}
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -s -k "reduce_applies_groupnorm" > $O/${P}_kern_rgn.log 2>&1; el "reduce+gn kernel tests exit $? : $(tail -1 $O/${P}_kern_rgn.log)"
grep -h "^\[reduce+gn\|^FAILED\|Error" $O/${P}_kern_rgn.log | cut -c1-200 | head -20
timeout 600 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -s > $O/${P}_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/${P}_unet.log)"
grep -h "^\[unet \|headroom\|^\[reduce+gn\|^FAILED" $O/${P}_unet.log | cut -c1-200 | head -40
timeout 900 python -m pytest tests/test_reference_script_gpu.py -q -p no:cacheprovider -s > $O/${P}_script.log 2>&1; el "reference script tests exit $? : $(tail -1 $O/${P}_script.log)"
grep -h "libsdmi calls\|^FAILED\|Error\|error" $O/${P}_script.log | cut -c1-300 | head -20
for r in 1 2; do
  SDMI_REDUCE_GN=0 timeout 300 python tools/unet_latency.py "GroupNorm-apply launches (REDUCE_GN=0)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  timeout 300 python tools/unet_latency.py "GroupNorm in the split-K reduction" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  [ -f $L/libsdmi_wt1.so ] && SDMI_LIB_PATH=$L/libsdmi_wt1.so timeout 300 python tools/unet_latency.py "sc1 fp32 16-byte stores (wt1)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  [ -f $L/libsdmi_wt2.so ] && SDMI_LIB_PATH=$L/libsdmi_wt2.so timeout 300 python tools/unet_latency.py "sc1 fp32 + fp16 stores (wt2)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_GN_APPLY_U4_QUADS=300000 timeout 300 python tools/unet_latency.py "gn_apply 4 quads/thread at 64x64" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_GN_APPLY_U4_QUADS=60000 timeout 300 python tools/unet_latency.py "gn_apply 4 quads/thread >= 32x32" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B"; cat $O/${P}_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/${P}_all.log 2>&1; el "whole GPU suite exit $? : $(tail -1 $O/${P}_all.log)"
grep -h "^FAILED\|^ERROR" $O/${P}_all.log | head
timeout 600 python bench.py --steps 6 --warmup 2 > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log | cut -c1-420
tail -1 $O/${P}_bench.log > $O/${P}_bench.json
SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes.txt 2>&1; el "per-shape table exit $?"; head -12 $O/${P}_shapes.txt
SDMI_REDUCE_GN=0 SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/${P}_shapes_off.txt 2>&1; el "per-shape table (REDUCE_GN=0) exit $?"; head -6 $O/${P}_shapes_off.txt
el done
