"""Which cold operand costs gn_conv3 its time?  384 MB are rewritten before every launch (nothing left in the L2s / Infinity Cache), then
x and / or the weights are read once by a copy kernel (warm again), then ONE launch between two events.
    python tools/bench_gn_conv3_cold.py [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import kernels as K  # noqa: E402
import test_gnconv_gpu as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, H, W = 2, 64, 64
c = T._case(B, H, W, 320, 0, 5)
N = c['N']
out = torch.empty(B * H * W, N, device='cuda')
o = K.groupnorm(c['x0'].view(B, H * W, -1), None, c['dgamma'], c['dbeta'], 1e-5, 1)
a16 = o['f16'].view(B * H * W, 320)
junk = torch.empty(96 << 20, device='cuda')
sx = torch.empty_like(c['x0']); sw = torch.empty_like(c['wp']); sa = torch.empty_like(a16); sr = torch.empty_like(c['dresid'])
import ctypes as C  # noqa: E402
from stable_diffusion_amd import _lib  # noqa: E402
lib = _lib.load()
n_ws = lib.sdmi_k_groupnorm_ws_floats(B, H * W)


def conv_only():
    K.igemm(a16, c['wp'], N, B, H, W, H, W, ksize=3, bias=c['dbias'], residual=c['dresid'], out_f32=out, splitk=1)


def fused():
    K.gn_conv3(c['x0'], None, c['dgamma'], c['dbeta'], 1e-5, c['wp'], N, out, bias=c['dbias'], residual=c['dresid'])


def timeit(fn, warm_x, warm_w, cold=True):
    for _ in range(3):
        fn()
    tot = 0.0
    for i in range(reps):
        if cold:
            junk.fill_(float(i))
        if warm_x:
            sx.copy_(c['x0']); sa.copy_(a16); sr.copy_(c['dresid'])
        if warm_w:
            sw.copy_(c['wp'])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / reps


for name, wx, ww, cold in (('hot', False, False, False), ('all cold', False, False, True), ('cold, activations re-read', True, False, True),
                           ('cold, weights re-read', False, True, True), ('cold, both re-read', True, True, True)):
    print(f'{name:28s}: conv {timeit(conv_only, wx, ww, cold):6.1f} us | stats + gn_conv3 {timeit(fused, wx, ww, cold):6.1f} us', flush=True)
