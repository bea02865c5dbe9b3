"""Hot vs cold-cache timing of a few igemm shapes: between timed launches a 1 GB buffer is rewritten so weights and
activations must come from HBM again (the situation inside a UNet call: 1.7 GB of weights stream through per call)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import kernels as K

SHAPES = [  # name, B, H, Cin, N, ksize
    ('L0 conv3 320->320', 2, 64, 320, 320, 3), ('L1 conv3 640->640', 2, 32, 640, 640, 3),
    ('L2 conv3 1280->1280', 2, 16, 1280, 1280, 3), ('L3 conv3 1280->1280', 2, 8, 1280, 1280, 3),
    ('L0 dense 320->320', 2, 64, 320, 320, 1), ('L0 dense 1280->320', 2, 64, 1280, 320, 1),
    ('L1 dense 640->640', 2, 32, 640, 640, 1), ('L2 dense 5120->1280', 2, 16, 5120, 1280, 1),
]
flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda')   # 1 GiB
g = torch.Generator().manual_seed(0)
for name, B, H, Cin, N, ks in SHAPES:
    x = torch.randn(B * H * H, Cin, generator=g).half().cuda()
    w = (torch.randn(N, ks * ks * Cin, generator=g) / math.sqrt(ks * ks * Cin)).half().cuda()
    out = torch.empty(B * H * H, N, device='cuda')
    fn = lambda: K.igemm(x, w, N, B, H, H, H, H, ks, 1, 0, out_f32=out, splitk=0)
    res = {}
    for mode in ('hot', 'cold'):
        ts = []
        for i in range(12):
            if os.environ.get('DBG'): print(name, mode, i, flush=True)
            if mode == 'cold':
                flush.fill_(float(i))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        res[mode] = ts[len(ts) // 2]
    print(f'{name:24s} hot {res["hot"]:7.1f} us   cold {res["cold"]:7.1f} us   x{res["cold"] / res["hot"]:.2f}', flush=True)
