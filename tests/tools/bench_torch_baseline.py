"""GPU baseline beside the CPU baseline (SURVEY.md 8c "On the GPU box"): the oracle's functional restatement of the
reference UNet (same aten ops as the reference modules: conv2d / linear / einsum-bmm / softmax / group_norm / layer_norm
/ gelu) on PyTorch-ROCm, fp32 and fp16 autocast, CFG batch 2 at 64x64 -- next to libsdmi on the same box.
    python tests/tools/bench_torch_baseline.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import unet_ref  # noqa: E402
from oracle.plan import SD_V1  # noqa: E402
from oracle.weights import make_inputs, make_state_dict  # noqa: E402

dev = torch.device('cuda')
sd = {k: v.to(dev) for k, v in make_state_dict(SD_V1, 0).items()}
x, t, ctx = (v.to(dev) for v in make_inputs(SD_V1, 2, 64, 64, seed=1))
orig_arange = torch.arange
torch.arange = lambda *a, **k: orig_arange(*a, **{**k, 'device': dev})     # the oracle builds the timestep table on the CPU


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = unet_ref.unet_forward(sd, SD_V1, x, t, ctx)
    ms32 = timeit(lambda: unet_ref.unet_forward(sd, SD_V1, x, t, ctx))
    with torch.autocast('cuda', dtype=torch.float16):
        ac = unet_ref.unet_forward(sd, SD_V1, x, t, ctx)
        ms16 = timeit(lambda: unet_ref.unet_forward(sd, SD_V1, x, t, ctx))
from stable_diffusion_amd import UNetModelHIP  # noqa: E402
m = UNetModelHIP(**SD_V1.ref_kwargs())
m.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)
m = m.cuda()
hip = m(x, t, context=ctx)
ms_hip = timeit(lambda: m(x, t, context=ctx), n=20)
print(f'UNet call, CFG batch 2, latent 64x64 (1606.5 GFLOP), same weights / inputs, one MI355X:')
print(f'  PyTorch-ROCm fp32            {ms32:8.3f} ms   {1606.5 / ms32:6.1f} TFLOP/s   (the truth the others are compared with)')
print(f'  PyTorch-ROCm fp16 autocast   {ms16:8.3f} ms   {1606.5 / ms16:6.1f} TFLOP/s   max-abs vs fp32 {(ac.float() - ref).abs().max().item():.3e}')
print(f'  libsdmi (this repo)          {ms_hip:8.3f} ms   {1606.5 / ms_hip:6.1f} TFLOP/s   max-abs vs fp32 {(hip - ref).abs().max().item():.3e}')
