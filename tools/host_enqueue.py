"""Host time to ENQUEUE one UNet call of the bench workload from an idle stream (no queue back-pressure: few calls, device idle at the start),
with and without the launch tapes:   python tools/host_enqueue.py [calls per sample = 2] [samples = 15]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

ncall = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 15
dev = torch.device('cuda:0')
ld, unet, vae = bench.build_gpu_model(dev)
g = torch.Generator(device='cpu').manual_seed(4)
x = torch.randn(2, 4, 64, 64, generator=g).to(dev)
ctx = (0.1 * torch.randn(2, 77, 768, generator=g)).to(dev)
t = torch.tensor([481, 481], device=dev)
unet.pin_context(ctx)
unet.cache_timesteps([481])
for rp in ('0', '1', '0', '1'):
    os.environ['SDMI_REPLAY'] = rp
    for _ in range(3):
        unet.hint_timestep(481)
        unet(x, t, context=ctx)
    hs, ds = [], []
    for _ in range(nsamp):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(ncall):
            unet.hint_timestep(481)
            unet(x, t, context=ctx)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append((t1 - t0) / ncall * 1e3)
        ds.append((t2 - t0) / ncall * 1e3)
    hs.sort(); ds.sort()
    print(f'SDMI_REPLAY={rp}: host enqueue median {hs[len(hs) // 2]:.3f} ms (min {hs[0]:.3f}) per UNet call of 319 launches; '
          f'enqueue + drain {ds[len(ds) // 2]:.3f} ms per call ({ncall} calls from an idle stream)', flush=True)
