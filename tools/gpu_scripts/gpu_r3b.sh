#!/bin/bash
# Round-3 GPU pass B: the GroupNorm-folding halo conv (conv3halo_gn_kernel): kernel tests (vs torch and bit-identity with the
# two-launch path), UNet goldens with it on, same-box A/B against SDMI_FUSE_GN_CONV=0, in-situ tuning of its (tile, split) keys
# only (SDMI_TUNE_ONLY_KSIZE=13), A/B and the whole GPU suite again with the tuned table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -p no:cacheprovider -k "gn_fold or conv3halo or groupnorm" > $O/b_kern.log 2>&1; el "kernel tests exit $? : $(tail -1 $O/b_kern.log)"
grep -h "gn-fold\|FAILED\|Error\|error" $O/b_kern.log | head -20
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -p no:cacheprovider > $O/b_unet.log 2>&1; el "unet tests exit $? : $(tail -1 $O/b_unet.log)"
grep -h "\[unet \|headroom\|TUNE_DISABLE" $O/b_unet.log | cut -c1-160 | head -20
for r in 1 2; do
  SDMI_FUSE_GN_CONV=0 timeout 300 python tools/unet_latency.py "FUSE_GN_CONV=0" 20 2 2>/dev/null | grep round >> $O/b_ab.txt
  SDMI_FUSE_GN_CONV=1 timeout 300 python tools/unet_latency.py "FUSE_GN_CONV=1 heuristic" 20 2 2>/dev/null | grep round >> $O/b_ab.txt
done
el "A/B (heuristic tiles)"; cat $O/b_ab.txt
cp stable-diffusion_amd/tune_gfx950.txt $O/b_tune.txt
SDMI_TUNE_ONLY_KSIZE=13 SDMI_TUNE_FILE=$O/b_tune.txt timeout 900 python tools/tune.py --workloads unet64,unet32,unet64b4,unet64b6,unet64b8 --rounds 40 --reps 3 --out $O/b_tune.txt --dump $O/b_tune_dump.txt > $O/b_tune.log 2>&1; el "tune exit $? : $(tail -1 $O/b_tune.log)"
grep " 13 1 0 0 " $O/b_tune.txt | head -40
for r in 1 2; do
  SDMI_FUSE_GN_CONV=0 timeout 300 python tools/unet_latency.py "FUSE_GN_CONV=0" 20 2 2>/dev/null | grep round >> $O/b_ab2.txt
  SDMI_TUNE_FILE=$O/b_tune.txt timeout 300 python tools/unet_latency.py "FUSE_GN_CONV=1 tuned" 20 2 2>/dev/null | grep round >> $O/b_ab2.txt
done
el "A/B (tuned)"; cat $O/b_ab2.txt
SDMI_TUNE_FILE=$O/b_tune.txt SDMI_PROF_SHAPES=1 timeout 300 python tools/prof_shapes.py > $O/b_shapes.txt 2>&1; el "prof_shapes exit $?"; grep -v amdgpu $O/b_shapes.txt | head -40
SDMI_TUNE_FILE=$O/b_tune.txt timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/b_pytest.log 2>&1; el "pytest -m gpu (tuned table) exit $? : $(tail -1 $O/b_pytest.log)"
grep -h "headroom" $O/b_pytest.log | cut -c1-200
SDMI_TUNE_FILE=$O/b_tune.txt timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/b_bench.log 2>&1; el "bench exit $?"; tail -1 $O/b_bench.log | cut -c1-300
el done
