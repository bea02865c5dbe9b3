"""Same-process, same-box A/B of run-time knobs on the UNet call of the bench workload (CFG batch 2, 64x64 latent, pinned context):
    python tools/unet_ab.py SDMI_REDUCE_GN_XCD=0 SDMI_REDUCE_GN_XCD=1 [--iters 20] [--rounds 4]
Each argument is one configuration (comma-separated NAME=VALUE pairs; 'base' = nothing set); the configurations are interleaved round by
round (A B A B ...) so that clock / thermal drift hits both.  Only knobs the library reads per call or per launch can be compared this way."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
iters = int(sys.argv[sys.argv.index('--iters') + 1]) if '--iters' in sys.argv else 20
rounds = int(sys.argv[sys.argv.index('--rounds') + 1]) if '--rounds' in sys.argv else 4
H = int(sys.argv[sys.argv.index('--latent') + 1]) if '--latent' in sys.argv else 64
args = [a for a in args if not a.isdigit()]
dev = torch.device('cuda:0')
ld, unet, vae = bench.build_gpu_model(dev)
res = {a: [] for a in args}
for r in range(rounds):
    for a in args:
        kv = [] if a == 'base' else [p.split('=') for p in a.split(',')]
        for k, v in kv:
            os.environ[k] = v
        res[a].append(bench.unet_latency_ms(unet, dev, H=H, W=H, iters=iters))
        for k, _ in kv:
            os.environ.pop(k)
for a in args:
    v = res[a]
    print(f'{a:40s} ' + ' / '.join(f'{x:.4f}' for x in v) + f'   median {sorted(v)[len(v) // 2]:.4f} ms per UNet call', flush=True)
