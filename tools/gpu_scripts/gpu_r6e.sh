cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r6e_pytest.log 2>&1; echo "pytest exit $? : $(tail -1 $O/r6e_pytest.log)"
grep "FAILED\|gelu_erf\] \|^\[unet headroom\|^.\[unet headroom" $O/r6e_pytest.log | cut -c1-300 | head -40
SDMI_LIB_PATH=$PWD/stable-diffusion_amd/libsdmi_exp.so timeout 900 python -m pytest tests -q -m "gpu and experiments" -p no:cacheprovider > $O/r6e_pytest_exp.log 2>&1; echo "experiments pytest exit $? : $(tail -1 $O/r6e_pytest_exp.log)"
grep "FAILED" $O/r6e_pytest_exp.log | head
