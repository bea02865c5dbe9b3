#!/bin/bash
# Round-2 GPU pass A: parity of the rewritten igemm (all tiles), the whole GPU suite, in-situ tuning, bench with / without
# the tuned table, per-shape profile.  Everything lands under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "igemm" -p no:cacheprovider -x > $O/a_igemm.log 2>&1; el "igemm tests exit $? : $(tail -1 $O/a_igemm.log)"
grep -E "FAILED|Error|error" $O/a_igemm.log | head -5
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_kernels_gpu.py::test_igemm_conv > $O/a_pytest.log 2>&1; el "gpu suite exit $? : $(tail -1 $O/a_pytest.log)"
grep -E "^FAILED|^ERROR" $O/a_pytest.log | head -20
grep -E "^\[unet " $O/a_pytest.log | head -20
timeout 900 python tools/tune.py --out $O/tune_gfx950.txt --dump $O/tune_dump.txt > $O/a_tune.log 2>&1; el "tune exit $? : $(tail -1 $O/a_tune.log)"
export SDMI_TUNE_FILE=$O/tune_gfx950.txt
timeout 900 python -m pytest tests/test_unet_gpu.py -q -p no:cacheprovider -k "golden" > $O/a_unet_tuned.log 2>&1; el "unet parity (tuned) exit $? : $(tail -1 $O/a_unet_tuned.log)"
grep -E "^\[unet " $O/a_unet_tuned.log | head -20
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/a_bench_tuned.log 2>&1; el "bench (tuned) exit $?"; tail -1 $O/a_bench_tuned.log | cut -c1-400
SDMI_TUNE_DISABLE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/a_bench_untuned.log 2>&1; el "bench (heuristic) exit $?"; tail -1 $O/a_bench_untuned.log | cut -c1-300
timeout 600 python tools/prof_shapes.py > $O/a_shapes_tuned.txt 2>&1; el "prof_shapes exit $?"; head -40 $O/a_shapes_tuned.txt
el done
