#!/bin/bash
# Round 6, pass G: launch tapes -- parity of replayed forwards, host enqueue time with / without them, then the whole suite with the tapes on.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6g}
O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 900 python -m pytest tests/test_tape_gpu.py -q -m gpu -p no:cacheprovider -s -x > $O/${P}_tape.log 2>&1; el "tape test exit $? : $(tail -1 $O/${P}_tape.log)"; grep "^\[tape\|^.\[tape\|Error\|assert" $O/${P}_tape.log | head -20
python - > $O/${P}_host.log 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device('cuda:0')
ld, unet, vae = bench.build_gpu_model(dev)
for r in range(3):
    for rp in ('0', '1'):
        os.environ['SDMI_REPLAY'] = rp
        ms = bench.unet_latency_ms(unet, dev, H=64, W=64, iters=20)
        print(f'SDMI_REPLAY={rp}  {ms:.4f} ms per UNet call, host enqueue {bench.unet_latency_ms.host_enqueue_ms:.3f} ms per call', flush=True)
PY
el "host exit $?"; grep SDMI_REPLAY $O/${P}_host.log
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/${P}_pytest.log 2>&1; el "pytest exit $? : $(tail -1 $O/${P}_pytest.log)"; grep "FAILED\|^E " $O/${P}_pytest.log | head
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/${P}_bench.log 2>&1; el "bench exit $?"; tail -1 $O/${P}_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','unet_ms_per_call','unet_host_enqueue_ms_per_call')})"
el done
