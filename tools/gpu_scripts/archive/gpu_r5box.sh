#!/bin/bash
# Round-5: the driver's bench command on one more box of the pool (box lottery record, with box_probe) and, once, the torchrun launch at N = 1 (RCCL init + all_gather)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r5box}
O=$PWD/gpurun_out; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/${P}_bench.log 2>&1; tail -1 $O/${P}_bench.log > $O/${P}_bench.json
python - "$P" <<'PY'
import json, sys
d = json.load(open(f'gpurun_out/{sys.argv[1]}_bench.json'))
print(f"images/s {d['value']:.3f}  ms/step {d['ms_per_step']:.1f}  unet ms/call {d['unet_ms_per_call']:.3f}  vae ms {d['vae_decode_ms']:.2f}  host enqueue ms/call {d.get('unet_host_enqueue_ms_per_call')}  probe {json.dumps(d['box_probe'])[:160]}")
PY
if [ "$2" = "rccl" ]; then
  NCCL_DEBUG=INFO timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_torchrun.log 2>&1
  echo "torchrun exit $?"; grep -i "nranks\|Init COMPLETE\|ranks_seen" $O/${P}_torchrun.log | cut -c1-200 | head -5; tail -1 $O/${P}_torchrun.log | cut -c1-300
fi
