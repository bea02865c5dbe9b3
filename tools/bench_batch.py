"""UNet latency vs CFG batch at 64x64 (the bench workload is B = 2: one prompt): how much of the MI355X a bigger batch
recovers.  python tools/bench_batch.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda')
ld, unet, vae = bench.build_gpu_model(dev)
print('UNet (SD-v1 architecture, random weights), latent 64x64, libsdmi; 1606.5 GFLOP per CFG pair')
for B in (2, 4, 6, 8):
    x = torch.randn(B, 4, 64, 64, device=dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.long)
    c = torch.randn(B, 77, 768, device=dev) * 0.1
    unet.pin_context(c)
    for _ in range(3):
        unet(x, t, context=c)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        unet(x, t, context=c)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    unet.unpin_context()
    print(f'  B={B} ({B // 2} prompt(s) with CFG): {ms:7.3f} ms/call  {ms / (B / 2):6.3f} ms per prompt-step  '
          f'{1606.5 * (B / 2) / ms:6.1f} TFLOP/s')
for B in (1, 4):
    lat = torch.randn(B, 4, 64, 64, device=dev) * 0.9
    bench.decode(vae, lat); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        bench.decode(vae, lat)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f'  first-stage decode B={B}: {ms:7.3f} ms  ({ms / B:6.3f} ms per image)')
