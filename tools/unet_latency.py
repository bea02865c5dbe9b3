"""UNet latency of the bench workload (CFG batch 2, 64x64 latent, pinned context) under the current environment / library:
    [SDMI_LIB_PATH=...] [SDMI_xxx=...] python tools/unet_latency.py [label] [iters] [rounds]
One line per round: label, ms per call.  For same-box A/Bs of env knobs and of two builds."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
label = sys.argv[1] if len(sys.argv) > 1 else 'unet'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
ld, unet, vae = bench.build_gpu_model(dev)
for r in range(rounds):
    ms = bench.unet_latency_ms(unet, dev, H=64, W=64, iters=iters)
    print(f'{label:28s} round {r}: {ms:.4f} ms per UNet call', flush=True)
