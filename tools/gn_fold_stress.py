#!/usr/bin/env python3
"""Stress / bisect harness of the GroupNorm-folding 3x3 convolution (csrc/conv3halo.hip conv3halo_gn_kernel).

The folding kernel must be BIT-identical to the two-launch path (GroupNorm-apply kernel -> LDS-DMA halo conv, same tile and
split): same fp16 operand, same MFMA sequence.  This script repeats that comparison many times per configuration at the sizes
of the UNet's 64x64 / 32x32 levels, with an L2 / Infinity-Cache flush in between (long memory latencies are what expose a wait
that is too weak), and for every mismatch prints WHERE it is: tile row, pixel row inside the tile, output-channel block.

    python tools/gn_fold_stress.py [--iters 20] [--flush 1] [--cases w64]

Runs on the GPU box only (tools/gpu_scripts/*.sh call it); exit code 1 when any iteration differed.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tests import kernels as K  # noqa: E402

DEV = 'cuda:0'

CASES = {
    'w64': [
        # name, B, H, W, c0, c1, N, splitk, tile
        ('w64_320_t14', 2, 64, 64, 320, 0, 320, 1, 14),
        ('w64_320_t14_k5', 2, 64, 64, 320, 0, 320, 5, 14),
        ('w64_320_t16', 2, 64, 64, 320, 0, 320, 1, 16),
        ('w64_320_t15', 2, 64, 64, 320, 0, 320, 1, 15),
        ('w64_320_t17', 2, 64, 64, 320, 0, 320, 1, 17),
        ('w64_640cat_t14_k2', 2, 64, 64, 320, 320, 320, 2, 14),
        ('w64_960cat_t14_k3', 2, 64, 64, 640, 320, 320, 3, 14),
        ('w64_960cat_t16', 2, 64, 64, 640, 320, 320, 1, 16),
    ],
    'w32': [
        ('w32_640_t14', 2, 32, 32, 640, 0, 640, 2, 14),
        ('w32_1280cat_t14', 2, 32, 32, 640, 640, 640, 4, 14),
        ('w32_960cat_t16', 2, 32, 32, 640, 320, 640, 3, 16),
    ],
    'w16': [
        ('w16_1280_t14', 2, 16, 16, 1280, 0, 1280, 5, 14),
        ('w16_2560cat_t14', 2, 16, 16, 1280, 1280, 1280, 10, 14),
    ],
}


def inputs(B, H, W, c0, c1, N, seed):
    g = torch.Generator().manual_seed(seed)
    C = c0 + c1
    x0 = torch.randn(B, H, W, c0, generator=g) * 1.3 + 0.2
    x1 = torch.randn(B, H, W, c1, generator=g) * 0.8 - 0.1 if c1 else None
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    w = torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    return x0, x1, gamma, beta, w, bias


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--flush', type=int, default=1)
    ap.add_argument('--cases', default='w64,w32,w16')
    ap.add_argument('--raw', type=int, default=1)
    a = ap.parse_args()
    flush = torch.empty(768 << 20, dtype=torch.uint8, device=DEV) if a.flush else None
    bad_total = 0
    for grp in a.cases.split(','):
        for name, B, H, W, c0, c1, N, splitk, tile in CASES[grp]:
            x0, x1, gamma, beta, w, bias = inputs(B, H, W, c0, c1, N, 7 + c0 + N)
            d = lambda t: None if t is None else t.to(DEV)
            x0d, x1d, gd, bd, wd, biasd = d(x0), d(x1), d(gamma), d(beta), d(w), d(bias)
            C = c0 + c1
            M = B * H * W
            gn = K.groupnorm(x0d.reshape(B, H * W, c0), None if x1d is None else x1d.reshape(B, H * W, c1), gd, bd, 1e-5, 1,
                             want=('f16',))
            ref = torch.full((M, N), float('nan'), device=DEV)
            K.igemm(gn['f16'].reshape(M, C), K.pack_conv_weight(wd), N, B, H, W, H, W, ksize=3, bias=biasd, out_f32=ref,
                    splitk=splitk, tile=tile, fused_splitk=False)
            torch.cuda.synchronize()
            nbad = 0
            for it in range(a.iters):
                if flush is not None:
                    flush.fill_(it & 255)
                out = K.conv3gn(x0d, x1d, gd, bd, 1e-5, wd, bias=biasd, splitk=splitk, tile=tile, want_raw=bool(a.raw))
                out = out[0] if a.raw else out
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    nbad += 1
                    diff = (out - ref).abs()
                    rows = (diff.amax(dim=1) > 0).nonzero().flatten()
                    cols = (diff.amax(dim=0) > 0).nonzero().flatten()
                    bm = 256 if tile in (14, 15) else 128
                    tiles = sorted(set((rows // bm).tolist()))
                    prow = sorted(set(((rows % bm) // W).tolist()))
                    print(f'  [{name}] it {it}: max {float(diff.max()):.3e}  rows {rows.numel()} (tiles {tiles[:12]}{"..." if len(tiles) > 12 else ""} '
                          f'rows-in-tile {prow})  cols {cols.numel()} [{int(cols.min())}..{int(cols.max())}]  nan {int(torch.isnan(out).sum())}',
                          flush=True)
            print(f'[{name}] {nbad} / {a.iters} iterations differ from the two-launch path', flush=True)
            bad_total += nbad
    print('TOTAL mismatching iterations:', bad_total)
    return 1 if bad_total else 0


if __name__ == '__main__':
    sys.exit(main())
