"""Micro-benchmark of every distinct GEMM / conv / attention shape of one SD-v1 UNet call (CFG batch 2) on the GPU,
sweeping the igemm tile / staging / split-K knobs.  Run on the GPU box; prints a table and writes gpurun_out/kernels.json.

    python tests/tools/bench_kernels.py [--h 64] [--quick]
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import kernels as K  # noqa: E402
from oracle.plan import SD_V1, build_plan  # noqa: E402

DEV = 'cuda'


def unet_gemm_shapes(H, B=2, L=77):
    """[(name, dict)] with multiplicities: every igemm launch of one UNet call."""
    plan = build_plan(SD_V1)
    shapes = {}

    def add(kind, **kw):
        key = (kind,) + tuple(sorted(kw.items()))
        shapes[key] = shapes.get(key, 0) + 1
    res = {}  # spatial size per block
    h = H
    hs = [h]
    cur = h
    def visit(L_, cur):
        if L_.kind == 'res':
            add('conv3', Cin=L_.cin, N=L_.cout, H=cur, stride=1, up=0)
            add('conv3', Cin=L_.cout, N=L_.cout, H=cur, stride=1, up=0)
            if L_.cin != L_.cout:
                add('dense', K=3 * L_.cin, N=L_.cout, M=B * cur * cur)
        elif L_.kind == 'attn':
            C, M = L_.cin, B * cur * cur
            add('dense', K=3 * C, N=C, M=M)       # proj_in (split3)
            add('dense', K=3 * C, N=C, M=M)       # proj_out
            add('heads', K=C, N=3 * C, M=M)
            add('dense', K=C, N=C, M=M)           # to_out 1
            add('heads', K=C, N=C, M=M)           # q2
            add('heads', K=768, N=2 * C, M=B * L)
            add('dense', K=C, N=C, M=M)           # to_out 2
            add('geglu', K=C, N=8 * C, M=M)
            add('dense', K=4 * C, N=C, M=M)
            add('attn', d=L_.d_head, nq=cur * cur, nkv=cur * cur)
            add('attn', d=L_.d_head, nq=cur * cur, nkv=L)
        elif L_.kind == 'down':
            add('conv3', Cin=L_.cin, N=L_.cout, H=cur, stride=2, up=0)
        elif L_.kind == 'up':
            add('conv3', Cin=L_.cin, N=L_.cout, H=cur, stride=1, up=1)
    for blk in plan.input_blocks[1:]:
        for L_ in blk:
            visit(L_, cur)
            if L_.kind == 'down':
                cur //= 2
    for L_ in plan.middle_block:
        visit(L_, cur)
    for blk in plan.output_blocks:
        for L_ in blk:
            visit(L_, cur)
            if L_.kind == 'up':
                cur *= 2
    return shapes


def time_fn(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--h', type=int, default=64)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--quick', action='store_true')
    args = ap.parse_args()
    B = 2
    shapes = unet_gemm_shapes(args.h, B)
    rows = []
    g = torch.Generator().manual_seed(0)
    total_best = 0.0
    for key, count in sorted(shapes.items(), key=lambda kv: str(kv[0])):
        kind = key[0]
        kw = dict(key[1:])
        if kind == 'attn':
            d, nq, nkv = kw['d'], kw['nq'], kw['nkv']
            heads = 8
            BH = B * heads
            q = torch.randn(BH, nq, d, generator=g).half().to(DEV)
            k = torch.randn(BH, nkv, d, generator=g).half().to(DEV)
            nkp = (nkv + 7) // 8 * 8
            vt = torch.randn(BH, d, nkp, generator=g).half().to(DEV)
            flops = 4.0 * BH * nq * nkv * d
            res = {}
            for nw in (2, 4, 8):
                os.environ['SDMI_ATTN_NW'] = str(nw)
                res[f'nw{nw}'] = time_fn(lambda: K.attention(q, k, vt, heads, nkv, d ** -0.5), args.iters)
            os.environ.pop('SDMI_ATTN_NW')
            res['auto'] = time_fn(lambda: K.attention(q, k, vt, heads, nkv, d ** -0.5), args.iters)
            best = min(res, key=res.get)
            ms = res[best]
            rows.append(dict(kind=kind, count=count, **kw, best=best, ms=ms, tflops=flops / ms / 1e9,
                             all={k_: round(v_, 4) for k_, v_ in res.items()}))
            total_best += ms * count
            continue
        if kind == 'conv3':
            Cin, N, H, stride, up = kw['Cin'], kw['N'], kw['H'], kw['stride'], kw['up']
            Hout = H * 2 if up else (H - 1) // stride + 1
            M = B * Hout * Hout
            a = torch.randn(B * H * H, Cin, generator=g).half().to(DEV)
            w = (torch.randn(N, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).half().to(DEV)
            geo = dict(B=B, Hin=H, Win=H, Hout=Hout, Wout=Hout, ksize=3, stride=stride, up=up)
            Kd = 9 * Cin
        else:
            M, Kd, N = kw['M'], kw['K'], kw['N']
            a = torch.randn(M, Kd, generator=g).half().to(DEV)
            w = (torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).half().to(DEV)
            geo = dict(B=1, Hin=M, Win=1, Hout=M, Wout=1, ksize=1, stride=1, up=0)
        flops = 2.0 * M * N * Kd
        out32 = torch.empty(M, N, device=DEV)
        out16 = torch.empty(M, max(N, 8), dtype=torch.float16, device=DEV)
        res = {}
        tiles = [0, 3] if kind == 'geglu' else [0, 1, 2, 3, 4, 5]
        splits = [1] if kind in ('geglu', 'heads') else ([1, 0] if args.quick else [1, 2, 4, 8, 16])
        for tile in tiles:
            for dma in [1]:
                for sk in splits:
                    if sk > 1 and (Kd // 64) // sk < 2:
                        continue
                    if kind == 'geglu':
                        fn = lambda: K.igemm(a, w, N, **geo, out_f16=out16[:, :N // 2], ldo=out16.stride(0), mode=1, tile=tile, dma=dma)
                    elif kind == 'heads':
                        C = N if N % 3 else N // 3
                        segs = N // C
                        C = N // segs
                        dh = C // 8
                        ntok = M // B
                        bufs = [torch.empty(M * C + 64, dtype=torch.float16, device=DEV) for _ in range(segs)]
                        hd = dict(segs=[(bufs[i], 1 if i == segs - 1 and segs > 1 else 0) for i in range(segs)], heads=8, dh=dh,
                                  ntok=ntok, ntok_pad=(ntok + 7) // 8 * 8, segC=C)
                        if hd['ntok_pad'] != ntok:
                            bufs[-1] = torch.empty(B * C * hd['ntok_pad'] + 64, dtype=torch.float16, device=DEV)
                            hd['segs'][-1] = (bufs[-1], 1)
                        fn = lambda: K.igemm(a, w, N, B, ntok, 1, ntok, 1, mode=2, heads=hd, tile=tile, dma=dma)
                    else:
                        fn = lambda: K.igemm(a, w, N, **geo, out_f32=out32, splitk=sk, tile=tile, dma=dma)
                    try:
                        ms = time_fn(fn, args.iters)
                    except Exception as e:  # noqa
                        ms = float('inf')
                    res[f't{tile}d{dma}k{sk}'] = ms
        best = min(res, key=res.get)
        row = dict(kind=kind, count=count)
        row.update(kw)
        row.update(M=M, Kd=Kd, best=best, ms=res[best], tflops=flops / res[best] / 1e9,
                   all={k: round(v, 4) for k, v in res.items()})
        rows.append(row)
        total_best += res[best] * count
    rows.sort(key=lambda r: -r['ms'] * r['count'])
    print(f'{"kind":6s} {"cnt":>3s} {"shape":44s} {"best":10s} {"ms":>8s} {"TF/s":>7s} {"tot ms":>7s}')
    for r in rows:
        shape = ' '.join(f'{k}={v}' for k, v in r.items() if k not in ('kind', 'count', 'best', 'ms', 'tflops', 'all'))
        print(f'{r["kind"]:6s} {r["count"]:3d} {shape:44s} {r["best"]:10s} {r["ms"]:8.4f} {r["tflops"]:7.1f} {r["ms"] * r["count"]:7.3f}')
    print(f'sum over launches with the best config per shape: {total_best:.3f} ms')
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'kernels.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
