#!/bin/bash
# Round 6, pass A: the multi-seed / realistic-statistics UNet goldens through the C ABI at the ONE 1e-3 bar, the gelu_erf pin,
# and a baseline bench line of the round-5 kernels on this box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6a}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 1500 python -m pytest tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet goldens exit $? : $(tail -1 $O/${P}_unet.log)"
grep "^\[unet" $O/${P}_unet.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -s -k "gelu_erf or geglu" > $O/${P}_gelu.log 2>&1; el "gelu exit $? : $(tail -1 $O/${P}_gelu.log)"; grep "gelu_erf\]" $O/${P}_gelu.log
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log > $O/${P}_bench.json; cut -c1-400 $O/${P}_bench.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r6a_bench.json'))
print({k: d.get(k) for k in ('value', 'unet_ms_per_call', 'unet_host_enqueue_ms_per_call', 'vae_decode_ms')})
print('box', d.get('box_probe'))
r = d['roofline']; print({k: r[k] for k in ('frac', 'frac_events', 'frac_scaled', 'avg_launch_ms', 'launches_per_unet_call', 'timing')})
print(d.get('cpu_baseline'))
PY
el done
