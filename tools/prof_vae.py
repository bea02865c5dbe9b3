"""Per-kernel-class timing of one first-stage decode / encode (HIP events per launch via sdmi_profile_begin/end)
and end-to-end latency next to the PyTorch-ROCm fp16-autocast decoder.   python tools/prof_vae.py [B]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stable_diffusion_amd import AutoencoderKLHIP, _lib  # noqa: E402
from stable_diffusion_amd.synthetic import SD_V1_VAE_DDCONFIG, randomize_vae_  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from vae_torch import AutoencoderKLDecoder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda')
vae = AutoencoderKLHIP(SD_V1_VAE_DDCONFIG, None, 4).to(dev).eval()
randomize_vae_(vae, 0)
lat = torch.randn(B, 4, 64, 64, device=dev) * 0.9
img = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
lib = _lib.load()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def profile(fn, title):
    fn(); torch.cuda.synchronize()
    _lib.check(lib.sdmi_profile_begin())
    fn(); torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 20)
    _lib.check(lib.sdmi_profile_end(buf, len(buf)))
    rows = sorted(json.loads(buf.value.decode()), key=lambda r: -r['ms'])
    tot = sum(r['ms'] for r in rows)
    print(f'== {title}: {tot:.3f} ms in {sum(r["launches"] for r in rows)} launches (serialised by the profiler)')
    for r in rows:
        tf = r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['flops'] else 0
        print(f'  {r["name"]:<28} n={r["launches"]:>3} {r["ms"]:8.3f} ms  {tf:7.1f} TF/s  {r["bytes"] / (r["ms"] * 1e-3) / 1e9:8.1f} GB/s')


print(f'B={B}: HIP decode {timeit(lambda: vae.decode_first_stage(lat)):.3f} ms, HIP encode {timeit(lambda: vae.encode_moments(img)):.3f} ms')
tv = AutoencoderKLDecoder().to(dev).eval()


def torch_dec():
    with torch.autocast('cuda', dtype=torch.float16):
        return tv.decode_first_stage(lat)


print(f'B={B}: PyTorch-ROCm fp16-autocast decode {timeit(torch_dec):.3f} ms')
profile(lambda: vae.decode_first_stage(lat), 'decode 64x64 -> 512x512')
profile(lambda: vae.encode_moments(img), 'encode 512x512 -> 64x64')
