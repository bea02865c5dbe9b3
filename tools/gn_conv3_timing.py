"""Where does gn_conv3_kernel spend its cycles INSIDE a UNet call (and stand-alone)?  Needs the timing build:
    bash tools/build_rc_timing.sh && SDMI_LIB_PATH=stable-diffusion_amd/libsdmi_rctiming.so python tools/gn_conv3_timing.py
Thread 0's cycle stamps of every workgroup: entry | tables (X0) | chunk 0 normalised | each chunk's 18 units | ring drained | epilogue | stores drained."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import bench  # noqa: E402
from stable_diffusion_amd import _lib  # noqa: E402

lib = _lib.load()
dbg_fn = getattr(lib, 'sdmi_k_gn_conv3_dbg')
dbg_fn.restype = C.c_int
dbg_fn.argtypes = [C.c_void_p]
dev = torch.device('cuda:0')
stamps = torch.zeros(16, 4096, 16, dtype=torch.int64, device=dev)


def report(tag, st, nwg):
    st = st[:nwg].double()
    d = st[:, 1:] - st[:, :-1]
    m = d.mean(0)
    names = ['tables', 'chunk0 norm', 'chunk 0', 'chunk 1', 'chunk 2', 'chunk 3', 'chunk 4', 'drain', 'epilogue', 'stores']
    tot = st[:, 10] - st[:, 0]
    start = st[:, 0] - st[:, 0].min()
    print(f'{tag}: workgroup cycles mean {tot.mean():8.0f} min {tot.min():8.0f} max {tot.max():8.0f} | entry skew mean {start.mean():7.0f} max {start.max():7.0f} | '
          f'first entry -> last end {(st[:, 10].max() - st[:, 0].min()):8.0f}', flush=True)
    print('    ' + ' | '.join(f'{n} {m[i]:6.0f}' for i, n in enumerate(names)), flush=True)


# stand-alone (hot)
import kernels as K  # noqa: E402
import test_gnconv_gpu as T  # noqa: E402
c = T._case(2, 64, 64, 320, 0, 5)
out = torch.empty(2 * 64 * 64, 320, device=dev)
for _ in range(3):
    K.gn_conv3(c['x0'], None, c['dgamma'], c['dbeta'], 1e-5, c['wp'], 320, out, bias=c['dbias'], residual=c['dresid'])
dbg_fn(stamps.data_ptr())
K.gn_conv3(c['x0'], None, c['dgamma'], c['dbeta'], 1e-5, c['wp'], 320, out, bias=c['dbias'], residual=c['dresid'])
torch.cuda.synchronize()
report('stand-alone, hot', stamps[0].cpu(), 256)

ld, unet, vae = bench.build_gpu_model(dev)
bench.unet_latency_ms(unet, dev, H=64, W=64, iters=3)
dbg_fn(stamps.data_ptr())
bench.unet_latency_ms(unet, dev, H=64, W=64, iters=1)       # (warm-up calls inside: the last call's seven launches are what remains)
torch.cuda.synchronize()
dbg_fn(None)
s = stamps.cpu()
for i in range(16):
    if s[i, 0, 0] != 0:
        report(f'in the UNet, launch slot {i:2d}', s[i], 256)
