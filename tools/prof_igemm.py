"""Launch one igemm shape repeatedly (for rocprofv3 kernel-trace / PMC passes).
    python tools/prof_igemm.py --shape l0conv --tile 3 --iters 30"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import kernels as K  # noqa: E402

SHAPES = {   # B, H, Cin, N, ksize
    'l0conv': (2, 64, 320, 320, 3),
    'l1conv': (2, 32, 640, 640, 3),
    'l2conv': (2, 16, 1280, 1280, 3),
    'l3conv': (2, 8, 1280, 1280, 3),
    'l0ff2': (2, 64, 1280, 320, 1),
    'big': (2, 64, 1280, 1280, 3),
    'geglu0': (2, 64, 320, 2560, 1),
    'geglu1': (2, 32, 640, 5120, 1),
    'dense0': (2, 64, 320, 320, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='l0conv')
    ap.add_argument('--tiles', default='3,5')
    ap.add_argument('--splitk', type=int, default=1)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    B, H, Cin, N, ks = SHAPES[a.shape]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B * H * H, Cin, generator=g).half().cuda()
    w = (torch.randn(N, ks * ks * Cin, generator=g) / math.sqrt(ks * ks * Cin)).half().cuda()
    out = torch.empty(B * H * H, N, device='cuda')
    geglu = a.shape.startswith('geglu')
    out16 = torch.empty(B * H * H, N // 2, dtype=torch.float16, device='cuda')
    bias = torch.zeros(N, device='cuda')
    resid = torch.randn(B * H * H, N, device='cuda') if not geglu else None
    for tile in [int(t) for t in a.tiles.split(',')]:
        for _ in range(a.iters):
            if geglu:
                K.igemm(x, w, N, B, H, H, H, H, ks, 1, 0, out_f16=out16, bias=bias, mode=1, tile=tile)
            else:
                K.igemm(x, w, N, B, H, H, H, H, ks, 1, 0, out_f32=out, bias=bias, residual=resid, tile=tile, splitk=a.splitk)
        torch.cuda.synchronize()
    print('done')


if __name__ == '__main__':
    main()
