"""Does pulling a GEMM's weights into the Infinity Cache ahead of time pay?  (round-3 experiment)

Per shape, HIP-event time of the GEMM launch (incl. its split-K reduce):
  hot          same operands as the previous launch (weights and activations in L2 / Infinity Cache)
  cold         1 GiB rewritten before every launch: weights AND activations come from HBM
  cold+A       cold, then the activation re-written by a copy kernel (what a real call looks like: the producer just wrote A)
  cold+A+W     ... and every 128-byte line of the weights touched by sdmi_k_prefetch_lines (one dword per line) before the launch
The difference (cold+A) - (cold+A+W) is what a weight prefetcher running one layer ahead could save at most.
"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import kernels as K
from stable_diffusion_amd import _lib

SHAPES = [  # name, B, H, Cin, N, ksize
    ('L0 conv3 320->320', 2, 64, 320, 320, 3), ('L1 conv3 640->640', 2, 32, 640, 640, 3),
    ('L2 conv3 1280->1280', 2, 16, 1280, 1280, 3), ('L3 conv3 1280->1280', 2, 8, 1280, 1280, 3),
    ('L0 dense 320->320', 2, 64, 320, 320, 1), ('L0 dense 1280->320', 2, 64, 1280, 320, 1),
    ('L1 dense 640->640', 2, 32, 640, 640, 1), ('L2 dense 5120->1280', 2, 16, 5120, 1280, 1),
    ('L3 dense 5120->1280', 2, 8, 5120, 1280, 1),
]
lib = _lib.load()
flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda')   # 1 GiB
g = torch.Generator().manual_seed(0)
for name, B, H, Cin, N, ks in SHAPES:
    x = torch.randn(B * H * H, Cin, generator=g).half().cuda()
    x_src = x.clone()
    w = (torch.randn(N, ks * ks * Cin, generator=g) / math.sqrt(ks * ks * Cin)).half().cuda()
    out = torch.empty(B * H * H, N, device='cuda')
    fn = lambda: K.igemm(x, w, N, B, H, H, H, H, ks, 1, 0, out_f32=out, splitk=0)
    res = {}
    for mode in ('hot', 'cold', 'cold+A', 'cold+A+W'):
        ts, tp = [], []
        for i in range(12):
            if mode != 'hot':
                flush.fill_(float(i))
            if mode in ('cold+A', 'cold+A+W'):
                x.copy_(x_src)
            torch.cuda.synchronize()
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if mode == 'cold+A+W':
                p0.record()
                _lib.check(lib.sdmi_k_prefetch_lines(w.data_ptr(), w.numel() * 2, _lib.stream_ptr()))
                p1.record()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
            if mode == 'cold+A+W':
                tp.append(p0.elapsed_time(p1) * 1e3)
        ts = sorted(ts[2:])
        res[mode] = ts[len(ts) // 2]
        if tp:
            res['touch'] = sorted(tp[2:])[len(tp[2:]) // 2]
    print(f'{name:24s} hot {res["hot"]:7.1f}  cold {res["cold"]:7.1f}  cold+A {res["cold+A"]:7.1f}  cold+A+W {res["cold+A+W"]:7.1f} us'
          f'   (touch kernel {res["touch"]:6.1f} us for {w.numel() * 2 / 1e6:5.1f} MB)', flush=True)
