"""GPU parity of every gfx950 kernel against fp32 torch restatements of the reference op (through the C ABI).

Operands are fp16-representable, so the reference value is exact up to fp32 accumulation order; tolerances are
written next to each check."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import kernels as K  # noqa: E402  (tests/ is on sys.path via conftest's rootdir insertion)

DEV = 'cuda'


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _rand16(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).half()


def _nhwc(x):  # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


ALL_TILES = list(range(14)) + [18, 19, 20, 21]     # include/sdmi.h: sdmi_igemm_desc.tile (generic implicit GEMM)
HALO_TILES = [14, 15, 16, 17]   # halo-staged 3x3 convolution: BM = 256, 256, 128, 128

CONV_CASES = [
    # name, B, Hin, Win, c0, c1, N, ksize, stride, up
    ('dense_masked', 2, 10, 10, 320, 0, 328, 1, 1, 0),
    ('conv3_s1', 2, 12, 12, 64, 0, 128, 3, 1, 0),
    ('conv3_s2', 2, 12, 12, 128, 0, 64, 3, 2, 0),
    ('conv3_s2_odd', 1, 7, 9, 64, 0, 64, 3, 2, 0),
    ('conv3_up', 2, 6, 6, 64, 0, 192, 3, 1, 1),
    ('conv3_cat', 2, 8, 8, 64, 128, 64, 3, 1, 0),
    ('conv1_cat', 2, 8, 8, 128, 64, 320, 1, 1, 0),
    ('tiny_m', 1, 2, 2, 256, 0, 256, 3, 1, 0),
    # long K (20 / 45 k-tiles through the 2..4-stage LDS-DMA ring), M and N tails inside the largest tiles
    ('dense_longk', 1, 300, 1, 1280, 0, 200, 1, 1, 0),
    ('conv3_longk', 1, 9, 11, 320, 0, 320, 3, 1, 0),
    ('conv3_up_odd', 1, 5, 7, 128, 0, 72, 3, 1, 1),
]


def _conv_ref(a0, a1, w, B, Hin, Win, ksize, stride, up):
    x = a0.float() if a1 is None else torch.cat([a0.float(), a1.float()], dim=1)
    C = x.shape[1]
    x = x.reshape(B, Hin, Win, C).permute(0, 3, 1, 2)
    if up:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    return F.conv2d(x, w.float(), None, stride=stride, padding=1 if ksize == 3 else 0)


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize('tile', ALL_TILES)
@pytest.mark.parametrize('dma', [0, 1])
def test_igemm_conv(case, tile, dma):
    name, B, Hin, Win, c0, c1, N, ksize, stride, up = case
    g = _g(hash(name) % 1000)
    Cin = c0 + c1
    # two channel-concatenated sources share one row pitch (they are column slices of one buffer, as in the executor)
    big = _rand16((B * Hin * Win, Cin), g)
    a0 = big[:, :c0]
    a1 = big[:, c0:] if c1 else None
    w = _rand16((N, Cin, ksize, ksize), g, 1.0 / math.sqrt(Cin * ksize * ksize))
    ref = _conv_ref(a0, a1, w, B, Hin, Win, ksize, stride, up)          # [B,N,Hout,Wout] fp32 (CPU)
    Hout, Wout = ref.shape[2], ref.shape[3]
    M = B * Hout * Wout
    bias = torch.randn(N, generator=g)
    rowvec = torch.randn(B, N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref2 = _nhwc(ref) + bias[None] + rowvec.repeat_interleave(Hout * Wout, dim=0) + resid
    wp = K.pack_conv_weight(w.float().to(DEV))
    out32 = torch.full((M, N), float('nan'), device=DEV)
    out16 = torch.full((M, N), float('nan'), device=DEV, dtype=torch.float16)
    big_d = big.to(DEV)
    K.igemm(big_d[:, :c0], wp, N, B, Hin, Win, Hout, Wout, ksize, stride, up, a1=big_d[:, c0:] if c1 else None,
            bias=bias.to(DEV), rowvec=rowvec.to(DEV), residual=resid.to(DEV), out_f32=out32, out_f16=out16,
            tile=tile, dma=dma)
    torch.cuda.synchronize()
    # fp32 accumulate of exact products: only summation order differs -> 2e-4 abs on O(1) outputs
    assert K.report(f'igemm {name} tile{tile} dma{dma} f32', out32, ref2, 2e-4) < 2e-4
    # fp16 copy: one rounding of |v| <= ~8 -> 2^-11 * 8 = 4e-3
    assert K.report(f'igemm {name} tile{tile} dma{dma} f16', out16, ref2, 6e-3) < 6e-3


@pytest.mark.parametrize('splitk', [0, 2, 5])
@pytest.mark.parametrize('tile', [0, 2, 3, 7, 8, 9, 11, 12, 13])
def test_igemm_splitk_inplace_residual(splitk, tile):
    g = _g(5)
    B, H, W, C, N = 2, 4, 4, 256, 192
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, 3, 3), g, 1.0 / math.sqrt(9 * C))
    bias = torch.randn(N, generator=g)
    resid = torch.randn(B * H * W, N, generator=g)
    ref = _nhwc(_conv_ref(a, None, w, B, H, W, 3, 1, 0)) + bias[None] + resid
    wp = K.pack_conv_weight(w.float().to(DEV))
    out = resid.clone().to(DEV)              # in place: out is also the residual (ResBlock skip path)
    out16 = torch.empty((B * H * W, N), dtype=torch.float16, device=DEV)
    K.igemm(a.to(DEV), wp, N, B, H, W, H, W, 3, 1, 0, bias=bias.to(DEV), residual=out, out_f32=out, out_f16=out16,
            splitk=splitk, tile=tile)
    torch.cuda.synchronize()
    assert K.report(f'igemm splitk{splitk} tile{tile}', out, ref, 2e-4) < 2e-4
    assert K.report(f'igemm splitk{splitk} tile{tile} f16', out16, ref, 6e-3) < 6e-3
    # split-K sums its slabs in a fixed order: bit-identical on a second run
    out2 = resid.clone().to(DEV)
    K.igemm(a.to(DEV), wp, N, B, H, W, H, W, 3, 1, 0, bias=bias.to(DEV), residual=out2, out_f32=out2, splitk=splitk, tile=tile)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


HALO_CASES = [
    # name, B, H, W, c0, c1, N, splitk
    ('h16', 2, 16, 16, 64, 0, 128, 1),
    ('h16_cat', 1, 16, 16, 64, 128, 72, 1),          # skip concat (two sources), N tail inside a tile
    ('h32', 1, 32, 32, 128, 0, 64, 1),
    ('h32_split', 2, 32, 32, 256, 0, 320, 2),        # split-K at 64-channel chunk granularity (4 chunks -> 2 + 2)
    ('h64', 2, 64, 64, 64, 0, 320, 1),               # the UNet's top level geometry (one chunk)
    ('h64_deep', 1, 64, 64, 320, 0, 64, 5),          # 5 chunks, one per split
    ('h16x32', 1, 16, 32, 192, 0, 128, 3),           # H != W
    ('h8_2img', 2, 8, 8, 128, 0, 64, 2),             # tiles of whole images (BM = 128: two 8x8 images, each with its own halo)
    ('h8_4img', 4, 8, 8, 64, 64, 128, 1),            # BM = 256: four images per tile
    ('h16_2img', 2, 16, 8, 64, 0, 96, 1),            # 16x8 images: BM = 128 is one image, BM = 256 two
]


@pytest.mark.parametrize('case', HALO_CASES, ids=[c[0] for c in HALO_CASES])
@pytest.mark.parametrize('tile', HALO_TILES)
def test_conv3halo(case, tile):
    """Halo-staged 3x3 conv (stride 1, pad 1; ResBlock convs, openaimodel.py:204,230) vs F.conv2d: every tile, image borders
    (zero padding = out-of-range buffer loads), chunk changes (double-buffered halo), split-K, and the GroupNorm statistics."""
    name, B, H, W, c0, c1, N, splitk = case
    bm = 256 if tile in (14, 15) else 128
    fits = (H * W) % bm == 0 if bm <= H * W else (bm % (H * W) == 0 and (B * H * W) % bm == 0)
    if not fits or bm % W:
        pytest.skip('tile rows do not fit this image')
    g = _g(hash(name) % 1000)
    Cin = c0 + c1
    big = _rand16((B * H * W, Cin), g)
    w = _rand16((N, Cin, 3, 3), g, 1.0 / math.sqrt(9 * Cin))
    ref = _conv_ref(big[:, :c0], big[:, c0:] if c1 else None, w, B, H, W, 3, 1, 0)
    M = B * H * W
    bias = torch.randn(N, generator=g)
    rowvec = torch.randn(B, N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref2 = _nhwc(ref) + bias[None] + rowvec.repeat_interleave(H * W, dim=0) + resid
    wp = K.pack_conv_weight(w.float().to(DEV))
    out32 = torch.full((M, N), float('nan'), device=DEV)
    big_d = big.to(DEV)
    gn = None
    if N % 32 == 0 and (H * W) % 32 == 0 and N // 32 >= 2:
        acc0 = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
        gn = [(acc0, N // 32, 0)]
    K.igemm(big_d[:, :c0], wp, N, B, H, W, H, W, 3, 1, 0, a1=big_d[:, c0:] if c1 else None, bias=bias.to(DEV),
            rowvec=rowvec.to(DEV), residual=resid.to(DEV), out_f32=out32, tile=tile, splitk=splitk, gn=gn)
    torch.cuda.synchronize()
    assert K.report(f'conv3halo {name} tile{tile} k{splitk}', out32, ref2, 3e-4) < 3e-4
    if gn:
        s_, ss_ = K.gn_acc_sums(acc0)
        v = out32.cpu().double().reshape(B, H * W, 32, N // 32)
        assert (s_ - v.sum((1, 3))).abs().max().item() < 2e-3
        assert ((ss_ - (v ** 2).sum((1, 3))).abs() / (1.0 + (v ** 2).sum((1, 3)))).max().item() < 1e-5
    # bit-identical on a second run (static schedule, fixed split order)
    out2 = torch.full((M, N), float('nan'), device=DEV)
    K.igemm(big_d[:, :c0], wp, N, B, H, W, H, W, 3, 1, 0, a1=big_d[:, c0:] if c1 else None, bias=bias.to(DEV),
            rowvec=rowvec.to(DEV), residual=resid.to(DEV), out_f32=out2, tile=tile, splitk=splitk)
    torch.cuda.synchronize()
    assert torch.equal(out32, out2)


@pytest.mark.parametrize('tile', [2, 5, 7, 8, 9, 12, 13])
@pytest.mark.parametrize('splitk', [1, 3])
@pytest.mark.parametrize('B,H,W,C,N', [(2, 8, 8, 128, 320), (3, 4, 8, 64, 200), (2, 16, 16, 64, 64)])
def test_igemm_groupnorm_statistics(tile, splitk, B, H, W, C, N):
    """GroupNorm(32) statistics of a conv output from the GEMM epilogue (split-K: from the reduce kernel): for the next
    layer's GroupNorm over the output alone (cpg = N / 32 ... here N / 20 to land group borders inside MFMA tiles) and for
    a skip-concat GroupNorm where the output is channels [cbase, cbase + N) (util.py:199-216, openaimodel.py:736)."""
    g = _g(91)
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, 3, 3), g, 1.0 / math.sqrt(9 * C))
    bias = torch.randn(N, generator=g)
    resid = torch.randn(B * H * W, N, generator=g)
    ref = _nhwc(_conv_ref(a, None, w, B, H, W, 3, 1, 0)) + bias[None] + resid        # [M, N]
    cpg0 = N // 20 if N % 20 == 0 else N // 8
    cbase1, cpg1 = 3 * cpg0 + 4, cpg0 + 4
    acc0 = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
    acc1 = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
    out = torch.full((B * H * W, N), float('nan'), device=DEV)
    K.igemm(a.to(DEV), K.pack_conv_weight(w.float().to(DEV)), N, B, H, W, H, W, 3, 1, 0, bias=bias.to(DEV),
            residual=resid.to(DEV), out_f32=out, splitk=splitk, tile=tile, gn=[(acc0, cpg0, 0), (acc1, cpg1, cbase1)])
    torch.cuda.synchronize()
    assert K.report(f'igemm+gn value tile{tile} k{splitk}', out, ref, 2e-4) < 2e-4
    v = out.cpu().double().reshape(B, H * W, N)
    for name, acc, cpg, cbase in (('next', acc0, cpg0, 0), ('concat', acc1, cpg1, cbase1)):
        s, ss = K.gn_acc_sums(acc)
        rs, rss = torch.zeros(B, 32, dtype=torch.float64), torch.zeros(B, 32, dtype=torch.float64)
        for n in range(N):
            gi = (cbase + n) // cpg
            rs[:, gi] += v[:, :, n].sum(1)
            rss[:, gi] += (v[:, :, n] ** 2).sum(1)
        e1, e2 = (s - rs).abs().max().item(), ((ss - rss).abs() / (1.0 + rss)).max().item()
        print(f'[gn-stats {name} tile{tile} k{splitk}] |sum err| {e1:.3e} rel sumsq err {e2:.3e}', flush=True)
        assert e1 < 2e-3 and e2 < 1e-5


@pytest.mark.parametrize('B,H,W,C,N,splitk,tile', [(2, 16, 16, 1280, 1280, 4, 2), (2, 8, 8, 1280, 1280, 15, 2),
                                                   (1, 5, 7, 256, 200, 3, 2), (2, 32, 32, 640, 640, 4, 3),
                                                   (1, 5, 7, 256, 200, 2, 0)])
@pytest.mark.parametrize('fused', [True, False])
def test_igemm_splitk_sd_shapes(B, H, W, C, N, splitk, tile, fused):
    """SD-sized split-K convs (hundreds of blocks on all XCDs write slabs; the last block of a tile -- fused -- or the
    reduce kernel sums them in split order): value, partial tiles, the per-batch row vector, 4 bit-identical repeats, and
    the two reductions agree to rounding."""
    g = _g(77)
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, 3, 3), g, 1.0 / math.sqrt(9 * C))
    bias = torch.randn(N, generator=g)
    rowvec = torch.randn(B, N, generator=g)
    resid = torch.randn(B * H * W, N, generator=g)
    ref = _nhwc(_conv_ref(a, None, w, B, H, W, 3, 1, 0)) + bias[None] + rowvec.repeat_interleave(H * W, dim=0) + resid
    wp = K.pack_conv_weight(w.float().to(DEV))
    a_d, bias_d, rv_d, res_d = a.to(DEV), bias.to(DEV), rowvec.to(DEV), resid.to(DEV)
    outs = []
    for rep in range(4):
        out = torch.full((B * H * W, N), float('nan'), device=DEV)
        K.igemm(a_d, wp, N, B, H, W, H, W, 3, 1, 0, bias=bias_d, rowvec=rv_d, residual=res_d, out_f32=out,
                splitk=splitk, tile=tile, fused_splitk=fused)
        outs.append(out)
    other = torch.full((B * H * W, N), float('nan'), device=DEV)
    K.igemm(a_d, wp, N, B, H, W, H, W, 3, 1, 0, bias=bias_d, rowvec=rv_d, residual=res_d, out_f32=other,
            splitk=splitk, tile=tile, fused_splitk=not fused)
    torch.cuda.synchronize()
    assert K.report(f'igemm splitk-sd splitk{splitk} tile{tile} M{B * H * W} N{N} fused={fused}', outs[0], ref, 3e-4) < 3e-4
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # (same split order; the epilogues add bias / row vector / residual in a different association: a few ulp)
    assert (other - outs[0]).abs().max().item() <= 4e-6 * ref.abs().max().item()


@pytest.mark.parametrize('B,H,W,C,N,splitk,tile,ksize,resid', [
    (2, 8, 8, 1280, 1280, 12, 19, 3, False),      # the 8x8 level: igemm 64x128, 12-way split (ResBlock conv1)
    (2, 16, 16, 1280, 1280, 10, 15, 3, False),    # 16x16: halo tile 256x128, 640 reduce workgroups
    (2, 16, 16, 640, 1280, 5, 14, 3, True),       # halo 256x64, with a residual
    (2, 16, 16, 1280, 640, 4, 6, 3, False),       # 20 channels per group: a wave's 32 columns touch three groups
    (1, 8, 8, 640, 512, 3, 5, 3, True),           # 16 channels per group (the narrowest accepted), one sample
])
def test_igemm_splitk_reduce_groupnorm_behind_a_grid_barrier(B, H, W, C, N, splitk, tile, ksize, resid, monkeypatch):
    """splitk_reduce_tiled_kernel<COOP> (SDMI_REDUCE_GN_XCD=1; round 6: behind the hierarchical grid barrier): the split-K reduction applies the consuming GroupNorm(32) + SiLU itself, behind a grid
    barrier between producing the statistics and using them (sdmi_igemm_desc::pgn_*; openaimodel.py:225-231 at the split levels).
    The statistics words must be the integers the plain reduction leaves (SDMI_REDUCE_GN_XCD=0), the fp16 output torch's GroupNorm + SiLU
    within one rounding, the library's own statistics + apply kernels up to rounding flips (their statistics partition the same values
    differently; the bit-identity with the two-launch path is checked on whole UNet calls), ten bit-identical repeats."""
    g = _g(917)
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, ksize, ksize), g, 1.0 / math.sqrt(ksize * ksize * C))
    bias = torch.randn(N, generator=g)
    rowvec = torch.randn(B, N, generator=g)
    res = torch.randn(B * H * W, N, generator=g) if resid else None
    gamma = 1.0 + 0.3 * torch.randn(N, generator=g)
    beta = 0.2 * torch.randn(N, generator=g)
    v_ref = _nhwc(_conv_ref(a, None, w, B, H, W, ksize, 1, 0)) + bias[None] + rowvec.repeat_interleave(H * W, dim=0)
    if resid:
        v_ref = v_ref + res
    y_ref = F.silu(F.group_norm(v_ref.view(B, H * W, N).permute(0, 2, 1).double(), 32, gamma.double(), beta.double(), 1e-5))
    y_ref = y_ref.permute(0, 2, 1).reshape(B * H * W, N).float()
    wp = K.pack_conv_weight(w.float().to(DEV))
    a_d, bias_d, rv_d = a.to(DEV), bias.to(DEV), rowvec.to(DEV)
    res_d = res.to(DEV) if resid else None
    ga_d, be_d = gamma.to(DEV), beta.to(DEV)

    def run(coop, keep):
        monkeypatch.setenv('SDMI_REDUCE_GN_XCD', coop)
        o16 = torch.full((B * H * W, N), float('nan'), dtype=torch.float16, device=DEV)
        o32 = torch.full((B * H * W, N), float('nan'), device=DEV)
        acc = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
        applied = K.igemm(a_d, wp, N, B, H, W, H, W, ksize, 1, 0, bias=bias_d, rowvec=rv_d, residual=res_d, out_f32=o32, splitk=splitk,
                          tile=tile, fused_splitk=True, pgn=(ga_d, be_d, 1e-5, 1, o16, keep), gn=[(acc, N // 32, 0)])
        torch.cuda.synchronize()
        return applied, o16, o32, acc

    ap0, _, v2, acc0 = run('0', 0)
    assert ap0 == 0 and not torch.isnan(v2).any()
    y2 = K.groupnorm(v2.view(B, H * W, N), None, ga_d, be_d, 1e-5, 1)['f16'].view(B * H * W, N)
    first = None
    for rep in range(10):
        ap, o16, o32, acc = run('1', rep % 2)
        assert ap == 1
        assert torch.equal(acc.sum(dim=2), acc0.sum(dim=2))   # the statistics the plain reduction leaves: the slot totals, integer for integer
        if rep % 2:
            assert torch.equal(o32, v2)                     # pgn_keep_f32: the reduction's fp32 value, bit for bit
        else:
            assert torch.isnan(o32).all()
        if first is None:
            first = o16
        assert torch.equal(o16, first), rep
    y = first.float().cpu()
    assert not torch.isnan(y).any()
    err = (y - y_ref).abs().max().item()
    print(f'[reduce+gn behind a grid barrier M{B * H * W} N{N} split{splitk} tile{tile}] max-abs vs torch {err:.3e}', flush=True)
    assert err <= 2e-3 * max(1.0, y_ref.abs().max().item())
    diff = (first.float() - y2.float()).abs()
    nflip = int((diff > 0).sum().item())
    assert diff.max().item() <= 8e-3 and nflip <= 2e-3 * diff.numel() + 2, (diff.max().item(), nflip)


@pytest.mark.parametrize('B,H,W,C,N,splitk,tile,ksize,resid', [
    (2, 8, 8, 1280, 1280, 12, 19, 3, False),      # the 8x8 level: igemm 64x128, 12-way split (ResBlock conv1)
    (2, 16, 16, 1280, 1280, 10, 15, 3, False),    # 16x16: halo tile, split at 64-channel chunks
    (2, 32, 32, 1280, 640, 5, 15, 3, False),      # 32x32 concat block: 20 channels per group, 5 quads per thread
    (2, 16, 16, 640, 1280, 5, 14, 3, True),       # with a residual
    (3, 8, 8, 256, 128, 2, 2, 1, False),          # smallest legal width: 4 channels per group
    (2, 16, 16, 320, 256, 0, -1, 3, False),       # auto split / auto tile: applied or not, the launcher says which
])
@pytest.mark.experiments
@pytest.mark.parametrize('tiled', ['1', '0'])
def test_igemm_splitk_reduce_applies_groupnorm(B, H, W, C, N, splitk, tile, ksize, resid, tiled, monkeypatch):
    """GroupNorm(32) + SiLU of a split-K conv's output applied by its reduction (splitk_reduce_gn_kernel, sdmi_igemm_desc::pgn_*):
    against torch's GroupNorm + SiLU of the fp32 conv result, four bit-identical repeats, and against this library's own
    GroupNorm kernels on the reduction's fp32 output (statistics kernel + apply: same elementwise function, statistics from a
    different partition of the same values -- rounding flips only; the bit-identity with the producers' fused statistics is
    checked on whole UNet calls, tests/test_unet_gpu.py)."""
    monkeypatch.setenv('SDMI_SLAB_TILED', tiled)      # register-order slabs (default) / row-major slabs: the kernel reads either
    monkeypatch.setenv('SDMI_REDUCE_GN', '1')         # (opt-in: measured slower than reduce + GroupNorm-apply)
    g = _g(91)
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, ksize, ksize), g, 1.0 / math.sqrt(ksize * ksize * C))
    bias = torch.randn(N, generator=g)
    rowvec = torch.randn(B, N, generator=g)
    res = torch.randn(B * H * W, N, generator=g) if resid else None
    gamma = 1.0 + 0.3 * torch.randn(N, generator=g)
    beta = 0.2 * torch.randn(N, generator=g)
    v_ref = _nhwc(_conv_ref(a, None, w, B, H, W, ksize, 1, 0)) + bias[None] + rowvec.repeat_interleave(H * W, dim=0)
    if resid:
        v_ref = v_ref + res
    y_ref = F.silu(F.group_norm(v_ref.view(B, H * W, N).permute(0, 2, 1).double(), 32, gamma.double(), beta.double(), 1e-5))
    y_ref = y_ref.permute(0, 2, 1).reshape(B * H * W, N).float()
    wp = K.pack_conv_weight(w.float().to(DEV))
    a_d, bias_d, rv_d = a.to(DEV), bias.to(DEV), rowvec.to(DEV)
    res_d = res.to(DEV) if resid else None
    ga_d, be_d = gamma.to(DEV), beta.to(DEV)
    outs = []
    for rep in range(4):
        o16 = torch.full((B * H * W, N), float('nan'), dtype=torch.float16, device=DEV)
        o32 = torch.full((B * H * W, N), float('nan'), device=DEV)
        # (the executor's situation: that GroupNorm is the output's one statistics target -- the condition under which the
        # reduction may apply it, because only then is the result the two-launch path's, bit for bit)
        acc = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
        applied = K.igemm(a_d, wp, N, B, H, W, H, W, ksize, 1, 0, bias=bias_d, rowvec=rv_d, residual=res_d, out_f32=o32,
                          splitk=splitk, tile=tile, fused_splitk=False, pgn=(ga_d, be_d, 1e-5, 1, o16, rep % 2),
                          gn=[(acc, N // 32, 0)])
        if applied:
            assert int(acc.abs().sum().item()) == 0        # no statistics atomics: the group never leaves the workgroup
        outs.append((applied, o16, o32))
    torch.cuda.synchronize()
    applied = outs[0][0]
    if splitk > 1:
        assert applied == 1
    # the library's two launches on the same inputs: the reduce kernel's fp32 output, then GroupNorm-apply
    v2 = torch.full((B * H * W, N), float('nan'), device=DEV)
    K.igemm(a_d, wp, N, B, H, W, H, W, ksize, 1, 0, bias=bias_d, rowvec=rv_d, residual=res_d, out_f32=v2, splitk=splitk, tile=tile,
            fused_splitk=False)
    y2 = K.groupnorm(v2.view(B, H * W, N), None, ga_d, be_d, 1e-5, 1)['f16'].view(B * H * W, N)
    torch.cuda.synchronize()
    if not applied:        # the auto choice did not split: nothing was applied, the fp32 output is the ordinary one
        assert torch.equal(outs[0][2], v2) and torch.isnan(outs[0][1].float()).all()
        return
    y = outs[0][1].float().cpu()
    err = (y - y_ref).abs().max().item()
    print(f'[reduce+gn M{B * H * W} N{N} split{splitk} tile{tile}] max-abs vs torch {err:.3e} (|y| max {y_ref.abs().max():.2f})', flush=True)
    assert err <= 2e-3 * max(1.0, y_ref.abs().max().item())          # one fp16 rounding of values up to ~4
    diff = (outs[0][1].float() - y2.float()).abs()
    nflip = int((diff > 0).sum().item())
    assert diff.max().item() <= 8e-3 and nflip <= 2e-3 * diff.numel() + 2, (diff.max().item(), nflip)    # rounding flips only
    for rep, (ap, o16, o32) in enumerate(outs):
        assert ap == 1 and torch.equal(o16, outs[0][1])
        if rep % 2:
            assert torch.equal(o32, v2)                 # pgn_keep_f32: the reduction's fp32 value, bit for bit
        else:
            assert torch.isnan(o32).all()               # ... and not written otherwise


@pytest.mark.parametrize('B,H,W,C,N,splitk,tile,ksize', [
    (2, 8, 8, 1280, 1280, 12, 19, 3),      # 8x8 level, igemm 64x128 deep ring
    (2, 16, 16, 1280, 1280, 10, 15, 3),    # halo tile 256x128, 8 waves
    (2, 16, 16, 640, 1280, 5, 14, 3),      # halo 256x64
    (2, 32, 32, 320, 320, 3, 10, 3),       # 64x64, 10 channels per group: quads straddle groups
    (1, 5, 7, 256, 200, 3, 2, 3),          # ragged M and N: padded tiles
    (3, 4, 8, 64, 200, 2, 8, 3),           # 64x128 tile, N not a tile multiple
    (2, 16, 16, 1280, 320, 4, 12, 1),      # 64x256 (1 x 4 waves), 1x1
    (2, 16, 16, 640, 640, 4, 13, 1),       # 256x64 (4 x 1 waves)
])
def test_splitk_register_order_slabs_are_bit_identical(B, H, W, C, N, splitk, tile, ksize, monkeypatch):
    """Unfused split-K with register-order slabs (sdmi::IGemmParams::slab_tiled: 16-byte write-through slab stores, 4 x 4 lane
    transposes in splitk_reduce_tiled_kernel; default) against the row-major slabs + splitk_reduce_kernel (SDMI_SLAB_TILED=0): the
    same partial sums added in the same order -- output, fp16 copy and GroupNorm statistics words must not differ by one bit."""
    g = _g(123)
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, ksize, ksize), g, 1.0 / math.sqrt(ksize * ksize * C))
    bias = torch.randn(N, generator=g)
    rowvec = torch.randn(B, N, generator=g)
    resid = torch.randn(B * H * W, N, generator=g)
    ref = _nhwc(_conv_ref(a, None, w, B, H, W, ksize, 1, 0)) + bias[None] + rowvec.repeat_interleave(H * W, dim=0) + resid
    wp = K.pack_conv_weight(w.float().to(DEV))
    a_d, bias_d, rv_d, res_d = a.to(DEV), bias.to(DEV), rowvec.to(DEV), resid.to(DEV)
    stats = (H * W) % 32 == 0
    cpg0 = N // 32 if N % 32 == 0 else max(2, N // 20)
    res = {}
    for tiled in ('1', '0'):
        monkeypatch.setenv('SDMI_SLAB_TILED', tiled)
        out = torch.full((B * H * W, N), float('nan'), device=DEV)
        o16 = torch.full((B * H * W, N), float('nan'), dtype=torch.float16, device=DEV)
        acc0 = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
        acc1 = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
        K.igemm(a_d, wp, N, B, H, W, H, W, ksize, 1, 0, bias=bias_d, rowvec=rv_d, residual=res_d, out_f32=out, out_f16=o16,
                splitk=splitk, tile=tile, fused_splitk=False,
                gn=[(acc0, cpg0, 0), (acc1, cpg0 + 4, 3 * cpg0 + 4)] if stats else None)
        torch.cuda.synchronize()
        res[tiled] = (out, o16, K.gn_acc_sums(acc0), K.gn_acc_sums(acc1))
    assert K.report(f'splitk tiled slabs tile{tile} M{B * H * W} N{N}', res['1'][0], ref, 3e-4) < 3e-4
    assert torch.equal(res['1'][0], res['0'][0]) and torch.equal(res['1'][1], res['0'][1])
    if stats:
        for t in (2, 3):
            assert torch.equal(res['1'][t][0], res['0'][t][0]) and torch.equal(res['1'][t][1], res['0'][t][1])


@pytest.mark.parametrize('ntok,d,heads,Kd,splitk,tile', [(256, 160, 8, 1280, 4, -1), (77, 64, 12, 768, 3, 5), (100, 40, 8, 1024, 2, 8)])
def test_splitk_register_order_slabs_head_scatter_bit_identical(ntok, d, heads, Kd, splitk, tile, monkeypatch):
    """... and the per-head scatter reduction (q / k rows, v^T columns; ragged token counts: padded tiles)."""
    g = _g(8)
    B = 2
    C = heads * d
    a = _rand16((B * ntok, Kd), g)
    w = _rand16((3 * C, Kd), g, 1.0 / math.sqrt(Kd))
    bias = (torch.randn(3 * C, generator=g) * 0.1).to(DEV)
    ntp = (ntok + 7) // 8 * 8
    res = {}
    for tiled in ('1', '0'):
        monkeypatch.setenv('SDMI_SLAB_TILED', tiled)
        q = torch.zeros((B * heads, ntok, d), dtype=torch.float16, device=DEV)
        k = torch.zeros_like(q)
        vt = torch.zeros((B * heads, d, ntp), dtype=torch.float16, device=DEV)
        K.igemm(a.to(DEV), w.to(DEV).contiguous(), 3 * C, B, ntok, 1, ntok, 1, mode=2, bias=bias, splitk=splitk, tile=tile, fused_splitk=False,
                heads=dict(segs=[(q, 0), (k, 0), (vt, 1)], heads=heads, dh=d, ntok=ntok, ntok_pad=ntp, segC=C))
        torch.cuda.synchronize()
        res[tiled] = (q, k, vt)
    y = (a.float() @ w.float().t() + bias.cpu()).reshape(B, ntok, 3, heads, d)
    assert K.report('tiled heads q', res['1'][0], y[:, :, 0].permute(0, 2, 1, 3).reshape(B * heads, ntok, d), 4e-3) < 4e-3
    for i in range(3):
        assert torch.equal(res['1'][i], res['0'][i])


def test_igemm_split_fp16_1x1():
    """3-pass split-fp16 1x1 conv (a_hi w_hi + a_lo w_hi + a_hi w_lo): fp32 operands to ~2^-22."""
    g = _g(9)
    M, Kd, N = 300, 320, 192
    x = torch.randn(M, Kd, generator=g) * 2.0
    w = torch.randn(N, Kd, generator=g) / math.sqrt(Kd)
    ref = (x.double() @ w.double().t()).float()
    hi, lo = K.cast_f16(x.to(DEV), want_lo=True)
    assert K.report('cast hi', hi, x, 8e-3) < 8e-3       # |x| up to ~10: one fp16 rounding
    assert K.report('cast hi+lo', hi.float() + lo.float(), x.to(DEV), 2e-6) < 2e-6
    wp = K.pack_split3(w.to(DEV))
    out = torch.empty((M, N), device=DEV)
    K.igemm(hi, wp, N, 1, M, 1, M, 1, a1=lo, a2=hi, out_f32=out)
    torch.cuda.synchronize()
    err3 = K.report('igemm split3', out, ref, 2e-5)
    out1 = torch.empty((M, N), device=DEV)
    K.igemm(hi, w.half().to(DEV).contiguous(), N, 1, M, 1, M, 1, out_f32=out1)
    err1 = K.report('igemm plain fp16 operands (for comparison)', out1, ref, 1e-2)
    assert err3 < 2e-5 and err3 < err1 / 20


@pytest.mark.parametrize('tile', [-1, 0, 1, 2, 4, 5, 8, 10])
@pytest.mark.parametrize('M,Kd,N,splitk', [(300, 320, 192, 1), (2048, 640, 640, 1), (512, 1280, 328, 4), (128, 1920, 1280, 6), (70, 64, 40, 1)])
def test_gemm_split16(tile, M, Kd, N, splitk):
    """split-fp16 dense GEMM family (gemm_split16.hip): a_hi w_hi + a_lo w_hi + a_hi w_lo from four operand tiles per 64-channel
    chunk -- fp32 operands to ~2^-22 -- against the fp64 product, against the K-concatenated formulation it replaces (same
    products, another summation order), with bias + residual, M / N tails and split-K; two runs are bit-identical."""
    g = _g(M + Kd + N)
    x = torch.randn(M, Kd, generator=g) * 2.0
    w = torch.randn(N, Kd, generator=g) / math.sqrt(Kd)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref = (x.double() @ w.double().t()).float() + bias[None] + resid
    hi, lo = K.cast_f16(x.to(DEV), want_lo=True)
    wp = K.pack_split3(w.to(DEV))
    out = torch.full((M, N), float('nan'), device=DEV)
    K.igemm(hi, wp, N, 1, M, 1, M, 1, a1=lo, bias=bias.to(DEV), residual=resid.to(DEV), out_f32=out, split16=True, tile=tile,
            splitk=splitk, fused_splitk=False)
    torch.cuda.synchronize()
    err = K.report(f'gemm_split16 M{M} K{Kd} N{N} tile{tile} k{splitk}', out, ref, 3e-5)
    old = torch.full((M, N), float('nan'), device=DEV)
    K.igemm(hi, wp, N, 1, M, 1, M, 1, a1=lo, a2=hi, bias=bias.to(DEV), residual=resid.to(DEV), out_f32=old)
    err_old = K.report('   ... K-concatenated formulation', old, ref, 3e-5)
    assert err < 3e-5 and err <= 2.0 * err_old + 2e-6
    out2 = torch.full((M, N), float('nan'), device=DEV)
    K.igemm(hi, wp, N, 1, M, 1, M, 1, a1=lo, bias=bias.to(DEV), residual=resid.to(DEV), out_f32=out2, split16=True, tile=tile,
            splitk=splitk, fused_splitk=False)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_igemm_geglu():
    """FeedForward/GEGLU, attention.py:37-64: value = first half, gate = second half, exact erf GELU."""
    g = _g(6)
    M, Kd, N = 200, 128, 512
    a = _rand16((M, Kd), g)
    w = _rand16((N, Kd), g, 1.0 / math.sqrt(Kd))
    b = torch.randn(N, generator=g) * 0.1
    y = a.float() @ w.float().t() + b
    val, gate = y.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    wp, bp = K.pack_geglu(w.float().to(DEV), b.to(DEV))
    for tile in (0, 3, 6, 7, 8, 9, 11, 12, 13):      # every tile whose waves own an even number of 32-column MFMA tiles
        out = torch.full((M, N // 2), float('nan'), device=DEV, dtype=torch.float16)
        K.igemm(a.to(DEV), wp, N, 1, M, 1, M, 1, bias=bp, out_f16=out, mode=1, tile=tile)
        torch.cuda.synchronize()
        assert K.report(f'igemm geglu tile{tile}', out, ref, 4e-3) < 4e-3


def test_gelu_erf_against_torch_erf():
    """`gelu_erf` (igemm_dev.h) is Abramowitz-Stegun 7.1.26 with the hardware reciprocal and a sign copy -- pin it
    against torch's exact-erf GELU (attention.py:43 `F.gelu`) THROUGH the GEGLU epilogue: one-hot operands make the accumulators
    exactly (1, g), so the fp16 output is fp16(gelu(g)).  Stated tolerance: at most TWO fp16 ulps from fp16(torch gelu(g)) anywhere on the
    gate range (measured: one ulp on 9 % of the gates, two only in the negative tail g < -4 where |gelu| < 1e-4 and the formula's 1.5e-7
    absolute error is a visible fraction of the value), the right sign everywhere including |g| -> 0 (where a copysign of a slightly
    negative erf would flip)."""
    M, Kd, N = 4096, 64, 128
    gate = torch.cat([torch.linspace(-8, 8, M - 64), torch.tensor([0.0, -0.0]), 10.0 ** -torch.arange(1, 32).float(),
                      -(10.0 ** -torch.arange(1, 32).float())]).half()
    a = torch.zeros((M, Kd), dtype=torch.float16)
    a[:, 0] = 1.0
    a[:, 1] = gate
    w = torch.zeros((N, Kd))
    w[:N // 2, 0] = 1.0           # value columns: 1
    w[N // 2:, 1] = 1.0           # gate columns: g
    wp, bp = K.pack_geglu(w.to(DEV), torch.zeros(N, device=DEV))
    out = torch.full((M, N // 2), float('nan'), device=DEV, dtype=torch.float16)
    K.igemm(a.to(DEV), wp, N, 1, M, 1, M, 1, bias=bp, out_f16=out, mode=1, tile=0)
    torch.cuda.synchronize()
    want = F.gelu(gate.double()).half()                  # exact erf in fp64, rounded once
    got = out[:, 0].cpu()
    assert torch.equal(out.cpu(), got[:, None].expand(-1, N // 2)), 'every column computes the same value'
    def ordered(h):          # sign-magnitude fp16 bits -> a monotonic integer (+0 and -0 coincide)
        b = h.view(torch.int16).int() & 0xffff
        return torch.where(b >= 0x8000, -(b & 0x7fff), b)
    ulps = (ordered(got) - ordered(want)).abs()
    same_sign = (got.float() * want.float() >= 0) | (got == 0) | (want == 0)
    worst = int(ulps.argmax())
    print(f'[gelu_erf] {int((ulps > 0).sum())} of {M} gates differ from fp16(torch erf gelu); {int((ulps > 1).sum())} by more than one fp16 ulp; '
          f'max {int(ulps.max())} ulp at gate {float(gate[worst]):.4g} (got {float(got[worst]):.6g}, want {float(want[worst]):.6g})', flush=True)
    assert bool(same_sign.all()) and int(ulps.max()) <= 2 and torch.isfinite(got).all()
    # in fp32 terms: |gelu_kernel - gelu_erf| <= 2 fp16 ulps of the value, i.e. a relative error <= 2^-10 -- the rounding of the fp16
    # output itself; the formula's own error (1.5e-7 absolute) is three orders of magnitude below it wherever |gelu| > 1e-3
    core = want.float().abs() > 1e-3
    assert int((ulps[core] > 1).sum()) <= int(0.002 * int(core.sum())) + 1


@pytest.mark.parametrize('ntok,d,heads', [(64, 40, 8), (77, 32, 2), (16, 160, 8), (100, 80, 4)])
def test_igemm_head_scatter(ntok, d, heads):
    """'b n (h d) -> (b h) n d' (attention.py:176) for q / k, and the transposed layout for v."""
    g = _g(7)
    B = 2
    C = heads * d
    Kd = 128
    M = B * ntok
    a = _rand16((M, Kd), g)
    w = _rand16((3 * C, Kd), g, 1.0 / math.sqrt(Kd))
    y = (a.float() @ w.float().t()).reshape(B, ntok, 3, heads, d)
    ntp = (ntok + 7) // 8 * 8
    q = torch.zeros((B * heads, ntok, d), dtype=torch.float16, device=DEV)
    k = torch.zeros_like(q)
    vt = torch.zeros((B * heads, d, ntp), dtype=torch.float16, device=DEV)
    wp = w.to(DEV).contiguous()
    K.igemm(a.to(DEV), wp, 3 * C, B, ntok, 1, ntok, 1, mode=2,
            heads=dict(segs=[(q, 0), (k, 0), (vt, 1)], heads=heads, dh=d, ntok=ntok, ntok_pad=ntp, segC=C))
    torch.cuda.synchronize()
    qr = y[:, :, 0].permute(0, 2, 1, 3).reshape(B * heads, ntok, d)
    kr = y[:, :, 1].permute(0, 2, 1, 3).reshape(B * heads, ntok, d)
    vr = y[:, :, 2].permute(0, 2, 3, 1).reshape(B * heads, d, ntok)
    assert K.report('heads q', q, qr, 4e-3) < 4e-3
    assert K.report('heads k', k, kr, 4e-3) < 4e-3
    assert K.report('heads vt', vt[:, :, :ntok], vr, 4e-3) < 4e-3
    assert float(vt[:, :, ntok:].abs().max()) == 0.0 if ntp != ntok else True


@pytest.mark.parametrize('ntok,d,heads,Kd,splitk', [(256, 160, 8, 1280, 4), (77, 64, 12, 768, 3), (100, 40, 8, 1024, 0)])
def test_igemm_head_scatter_splitk(ntok, d, heads, Kd, splitk):
    """per-head scatter with split-K: slabs + splitk_reduce_heads_kernel (bias included, ragged token counts)."""
    g = _g(8)
    B = 2
    C = heads * d
    M = B * ntok
    a = _rand16((M, Kd), g)
    w = _rand16((3 * C, Kd), g, 1.0 / math.sqrt(Kd))
    bias = torch.randn(3 * C, generator=g) * 0.1
    y = (a.float() @ w.float().t() + bias).reshape(B, ntok, 3, heads, d)
    ntp = (ntok + 7) // 8 * 8
    outs = []
    for rep in range(2):
        q = torch.zeros((B * heads, ntok, d), dtype=torch.float16, device=DEV)
        k = torch.zeros_like(q)
        vt = torch.zeros((B * heads, d, ntp), dtype=torch.float16, device=DEV)
        K.igemm(a.to(DEV), w.to(DEV).contiguous(), 3 * C, B, ntok, 1, ntok, 1, mode=2, bias=bias.to(DEV), splitk=splitk,
                heads=dict(segs=[(q, 0), (k, 0), (vt, 1)], heads=heads, dh=d, ntok=ntok, ntok_pad=ntp, segC=C))
        outs.append((q, k, vt))
    torch.cuda.synchronize()
    q, k, vt = outs[0]
    qr = y[:, :, 0].permute(0, 2, 1, 3).reshape(B * heads, ntok, d)
    kr = y[:, :, 1].permute(0, 2, 1, 3).reshape(B * heads, ntok, d)
    vr = y[:, :, 2].permute(0, 2, 3, 1).reshape(B * heads, d, ntok)
    assert K.report(f'heads splitk{splitk} q', q, qr, 6e-3) < 6e-3
    assert K.report(f'heads splitk{splitk} k', k, kr, 6e-3) < 6e-3
    assert K.report(f'heads splitk{splitk} vt', vt[:, :, :ntok], vr, 6e-3) < 6e-3
    assert all(torch.equal(x, y_) for x, y_ in zip(outs[0], outs[1]))


@pytest.mark.parametrize('tile', [0, 3, 5, 6, 7, 8, 9, 10, 12, 13, 14, 15, 16, 17])
def test_igemm_16_byte_epilogue_bit_identical(tile, monkeypatch):
    """16-byte epilogues (accumulators turned through LDS, igemm.hip) against the dword / short ones (SDMI_EPI_VEC=0), same
    launch otherwise: plain mode with bias + row vector + residual + fp32 and fp16 outputs + GroupNorm statistics, split-K
    slabs, GEGLU and the per-head scatter -- every output bit for bit."""
    g = _g(123)
    B, H, W, C, N = 2, 16, 16, 128, 256
    M = B * H * W
    a = _rand16((M, C), g).to(DEV)
    w = _rand16((N, C, 3, 3), g, 1.0 / math.sqrt(9 * C))
    wp = K.pack_conv_weight(w.float().to(DEV))
    bias, rowvec, resid = torch.randn(N, generator=g).to(DEV), torch.randn(B, N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    halo = 14 <= tile <= 17

    def run_all():
        res = []
        for splitk in (1, 2):
            o32 = torch.full((M, N), float('nan'), device=DEV)
            o16 = torch.zeros((M, N), dtype=torch.float16, device=DEV)
            acc0 = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
            K.igemm(a, wp, N, B, H, W, H, W, 3, 1, 0, bias=bias, rowvec=rowvec, residual=resid, out_f32=o32, out_f16=o16,
                    splitk=splitk, tile=tile, fused_splitk=False, gn=[(acc0, N // 32, 0)])
            res += [o32, o16, acc0]
        if not halo:
            wl = _rand16((N, C), g0, 1.0 / math.sqrt(C)).to(DEV)
            if tile in (0, 3, 6, 7, 8, 9, 12, 13):                    # GEGLU: waves own an even number of 32-column tiles
                wg, bg = K.pack_geglu(wl.float(), bias)
                og = torch.zeros((M, N // 2), dtype=torch.float16, device=DEV)
                K.igemm(a, wg, N, 1, M, 1, M, 1, bias=bg, out_f16=og, mode=1, tile=tile)
                res.append(og)
            heads, d = 2, 32                                          # q | k | v^T, 3 x 64 columns = 192 of the 256 weight rows
            ntok = H * W
            q = torch.zeros((B * heads, ntok, d), dtype=torch.float16, device=DEV)
            k = torch.zeros_like(q)
            vt = torch.zeros((B * heads, d, ntok), dtype=torch.float16, device=DEV)
            K.igemm(a, wl[:192].contiguous(), 192, B, ntok, 1, ntok, 1, mode=2, tile=tile, bias=bias[:192].contiguous(),
                    heads=dict(segs=[(q, 0), (k, 0), (vt, 1)], heads=heads, dh=d, ntok=ntok, ntok_pad=ntok, segC=heads * d))
            res += [q, k, vt]
        torch.cuda.synchronize()
        return res

    g0 = _g(5)
    monkeypatch.setenv('SDMI_EPI_VEC', '1')
    r1 = run_all()
    g0 = _g(5)
    monkeypatch.setenv('SDMI_EPI_VEC', '0')
    r0 = run_all()
    assert len(r0) == len(r1) and all(torch.isfinite(x.float()).all() for x in r1)
    for i, (x, y) in enumerate(zip(r1, r0)):
        assert torch.equal(x, y), (i, float((x.float() - y.float()).abs().max()))


def test_igemm_sd_l0_conv_shape():
    """The dominant SD-v1 shape: 320->320 3x3 at 64x64, CFG batch 2."""
    g = _g(8)
    B, H, W, C, N = 2, 64, 64, 320, 320
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, 3, 3), g, 1.0 / math.sqrt(9 * C))
    wp = K.pack_conv_weight(w.float().to(DEV))
    out = torch.empty((B * H * W, N), device=DEV)
    K.igemm(a.to(DEV), wp, N, B, H, W, H, W, 3, 1, 0, out_f32=out)
    torch.cuda.synchronize()
    x = a.float().to(DEV).reshape(B, H, W, C).permute(0, 3, 1, 2)
    ref = _nhwc(F.conv2d(x, w.float().to(DEV), None, padding=1))
    assert K.report('igemm L0 conv', out, ref, 3e-4) < 3e-4


ATTN_CASES = [
    # d, heads, nq, nkv
    (40, 8, 256, 256), (40, 8, 4096, 4096), (80, 8, 1024, 1024), (160, 8, 256, 256), (160, 8, 64, 64),
    (40, 8, 256, 77), (80, 8, 64, 77), (160, 8, 64, 77), (32, 2, 16, 16), (64, 2, 4, 4), (128, 2, 100, 77),
    (40, 2, 200, 130), (64, 2, 96, 96),
    # >= 512 keys: the software-pipelined kernel (odd / even tile counts, ragged last tile, every head dim it serves)
    (40, 8, 600, 577), (80, 4, 300, 512), (64, 2, 130, 1000), (32, 2, 64, 640), (40, 2, 2304, 2304), (80, 2, 100, 1016),
    (40, 2, 96, 520), (160, 2, 576, 576), (128, 2, 64, 777),
]


@pytest.mark.parametrize('d,heads,nq,nkv', ATTN_CASES)
def test_attention(d, heads, nq, nkv):
    """CrossAttention core, attention.py:178-192: softmax(q k^T * d^-0.5) v, heads merged 'b n (h d)'."""
    g = _g(d + nq + nkv)
    B = 2 if nq < 4096 else 1
    BH = B * heads
    q = _rand16((BH, nq, d), g)
    k = _rand16((BH, nkv, d), g)
    v = _rand16((BH, nkv, d), g)
    # a few large scores so the online-softmax rescale path is exercised (guide rule 26)
    q[0, 0] *= 6.0
    k[0, nkv - 1] *= 6.0
    k[0, nkv // 2] *= 5.0
    scale = d ** -0.5
    sim = torch.bmm(q.float(), k.float().transpose(1, 2)) * scale
    ref = torch.bmm(sim.softmax(-1), v.float())
    ref = ref.reshape(B, heads, nq, d).permute(0, 2, 1, 3).reshape(B, nq, heads * d)
    nkp = (nkv + 7) // 8 * 8
    vt = torch.zeros((BH, d, nkp), dtype=torch.float16)
    vt[:, :, :nkv] = v.transpose(1, 2)
    out = K.attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads, nkv, scale)
    torch.cuda.synchronize()
    # P is rounded to fp16 before PV (rel 2^-11) and the output is stored as fp16: 3e-3 abs on |v| ~ N(0,1)
    assert K.report(f'attention d{d} nq{nq} nkv{nkv}', out, ref, 3e-3) < 3e-3


@pytest.mark.parametrize('heads,nq,nkv', [(8, 4096, 4096), (2, 2048, 4096), (3, 4096 + 64, 2048 + 128), (1, 9216, 9216)])
def test_attention_key_split(heads, nq, nkv, monkeypatch):
    """attn_dma_kernel KVS = 2 (round 6; d = 40 self-attention of the 64 x 64 / 96 x 96 levels): two 8-wave groups of one workgroup take half of the
    keys each and merge their softmax partials through LDS.  Against fp32 torch like test_attention (same tolerance), against the one-group
    kernel (the two differ by the fp32 rounding of the merge and by P being rounded relative to each half's running maximum), with outliers in
    BOTH halves and at the seam so that both rescale paths and the merge weights 2^(m_g - m) run; ragged query counts; bit-identical repeats."""
    d = 40
    g = _g(7000 + nq + nkv)
    BH = heads
    q = _rand16((BH, nq, d), g)
    k = _rand16((BH, nkv, d), g)
    v = _rand16((BH, nkv, d), g)
    q[0, 0] *= 6.0
    k[0, nkv - 1] *= 6.0                   # second half
    k[0, nkv // 2 - 1] *= 5.0              # last key of the first half
    k[0, 3] *= 7.0                         # first half
    scale = d ** -0.5
    ref = torch.cat([torch.bmm((torch.bmm(q[i:i + 1].float(), k[i:i + 1].float().transpose(1, 2)) * scale).softmax(-1), v[i:i + 1].float())
                     for i in range(BH)]).reshape(1, heads, nq, d).permute(0, 2, 1, 3).reshape(1, nq, heads * d)
    vt = v.transpose(1, 2).contiguous()
    qd, kd, vd = q.to(DEV), k.to(DEV), vt.to(DEV)
    monkeypatch.setenv('SDMI_ATTN_KVS', '0')
    one = K.attention(qd, kd, vd, heads, nkv, scale).clone()
    monkeypatch.setenv('SDMI_ATTN_KVS', '2')          # (2 = also above 4096 keys, where the default keeps the 8-wave kernel)
    two = K.attention(qd, kd, vd, heads, nkv, scale).clone()
    again = K.attention(qd, kd, vd, heads, nkv, scale)
    torch.cuda.synchronize()
    assert torch.equal(two, again)
    from stable_diffusion_amd import _lib
    if _lib.load().sdmi_has_experiments():
        # experiments build: the rotated second key group and one barrier per two key tiles -- the same values, the same bits (both measured slower)
        monkeypatch.setenv('SDMI_ATTN_ROT', '1')
        rot = K.attention(qd, kd, vd, heads, nkv, scale).clone()
        monkeypatch.setenv('SDMI_ATTN_ROT', '0')
        monkeypatch.setenv('SDMI_ATTN_TPB', '2')
        tp = K.attention(qd, kd, vd, heads, nkv, scale).clone()
        monkeypatch.setenv('SDMI_ATTN_KVS', '0')
        tp8 = K.attention(qd, kd, vd, heads, nkv, scale).clone()
        torch.cuda.synchronize()
        assert torch.equal(rot, two) and torch.equal(tp, two) and torch.equal(tp8, one)
    e1 = K.report(f'attention one group  d40 nq{nq} nkv{nkv}', one, ref, 3e-3)
    e2 = K.report(f'attention key split  d40 nq{nq} nkv{nkv}', two, ref, 3e-3)
    dd = (one.float() - two.float()).abs().max().item()
    print(f'[attention key split nq{nq} nkv{nkv}] split vs one group max-abs {dd:.3e}', flush=True)
    assert e2 < 3e-3 and e2 <= 1.5 * e1 + 2e-4 and dd < 3e-3


@pytest.mark.parametrize('d,heads,n', [(64, 12, 77), (64, 2, 40), (32, 4, 130)])
def test_attention_causal(d, heads, n):
    """CLIPTextModel self-attention: softmax over keys <= query (transformers modeling_clip.py, causal mask)."""
    g = _g(41)
    B = 2
    q = _rand16((B * heads, n, d), g); k = _rand16((B * heads, n, d), g); v = _rand16((B * heads, n, d), g)
    scale = d ** -0.5
    mask = torch.full((n, n), float('-inf')).triu(1)
    p = torch.softmax(q.float() @ k.float().transpose(1, 2) * scale + mask, dim=-1)
    ref = (p @ v.float()).reshape(B, heads, n, d).permute(0, 2, 1, 3).reshape(B, n, heads * d)
    npad = (n + 7) // 8 * 8
    vt = torch.zeros((B * heads, d, npad), dtype=torch.float16)
    vt[:, :, :n] = v.transpose(1, 2)
    out = K.attention_causal(q.to(DEV), k.to(DEV), vt.to(DEV), heads, scale)
    assert K.report(f'attention causal d{d} n{n}', out, ref, 4e-3) < 4e-3


@pytest.mark.parametrize('c0,c1,HW,silu,eps', [(320, 0, 64 * 64, 1, 1e-5), (1280, 640, 16 * 16, 1, 1e-5),
                                               (64, 0, 4, 0, 1e-6), (640, 320, 100, 1, 1e-5), (2560, 0, 64, 1, 1e-5)])
def test_groupnorm(c0, c1, HW, silu, eps):
    """GroupNorm32 + SiLU (util.py:199-216, openaimodel.py:201-203) over the channel concat of two sources."""
    g = _g(c0 + HW)
    B = 2
    C = c0 + c1
    x0 = torch.randn(B, HW, c0, generator=g) * 1.5 + 0.3
    x1 = torch.randn(B, HW, c1, generator=g) * 0.7 - 0.2 if c1 else None
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    x = x0 if x1 is None else torch.cat([x0, x1], dim=2)
    ref = F.group_norm(x.permute(0, 2, 1).reshape(B, C, HW, 1), 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.reshape(B, C, HW).permute(0, 2, 1)
    o = K.groupnorm(x0.to(DEV), None if x1 is None else x1.to(DEV), gamma.to(DEV), beta.to(DEV), eps, silu,
                    want=('f16', 'f32', 'raw', 'lo', 'raw_lo'))
    torch.cuda.synchronize()
    assert K.report('groupnorm f32', o['f32'], ref, 2e-5) < 2e-5
    assert K.report('groupnorm f16', o['f16'], ref, 4e-3) < 4e-3
    assert K.report('groupnorm raw', o['raw'], x, 4e-3) < 4e-3
    assert K.report('groupnorm hi+lo', o['f16'].float() + o['lo'].float(), o['f32'], 4e-6) < 4e-6
    assert K.report('groupnorm raw hi+lo', o['raw'].float() + o['raw_lo'].float(), x, 4e-6) < 4e-6


@pytest.mark.parametrize('M,C', [(8192, 320), (512, 1280), (7, 64), (100, 640), (2048, 640), (33, 512), (5, 768), (9, 772)])
def test_layernorm(M, C, monkeypatch):
    g = _g(M + C)
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    monkeypatch.setenv('SDMI_LN_SLOTS', '1')
    out = K.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV)).clone()
    monkeypatch.setenv('SDMI_LN_SLOTS', '0')         # the 5-slot instantiation for every width (the only one before)
    out5 = K.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV)).clone()
    torch.cuda.synchronize()
    assert K.report('layernorm', out, ref, 4e-3) < 4e-3
    assert torch.equal(out, out5)


def test_time_embedding_path():
    """timestep_embedding (util.py:151-171) + time_embed MLP + emb_layers (openaimodel.py:506-511,218-224), fp32."""
    from oracle.unet_ref import timestep_embedding as ref_temb
    g = _g(11)
    t = torch.tensor([981, 481, 1, 0, 999], dtype=torch.int64)
    ref = ref_temb(t, 320)
    out = K.timestep_embedding(t.to(DEV), 320)
    torch.cuda.synchronize()
    assert K.report('timestep_embedding', out, ref, 2e-4) < 2e-4     # arg up to 999 rad: fp32 range reduction
    outf = K.timestep_embedding(t.float().to(DEV), 320)
    assert K.report('timestep_embedding(float t)', outf, ref, 2e-4) < 2e-4
    w = torch.randn(1280, 320, generator=g) / math.sqrt(320)
    b = torch.randn(1280, generator=g) * 0.1
    x = torch.randn(5, 320, generator=g)
    for silu in (0, 1):
        refl = F.linear(F.silu(x) if silu else x, w, b)
        outl = K.small_linear(x.to(DEV), w.to(DEV), b.to(DEV), silu)
        assert K.report(f'small_linear silu{silu}', outl, refl, 2e-5) < 2e-5


@pytest.mark.parametrize('tile', [0, 2, 3, 7, 9, 12, 13])
@pytest.mark.parametrize('B,H,W,C,N', [(2, 12, 12, 64, 128), (1, 6, 10, 128, 64)])
def test_igemm_conv_stride2_asym_pad(tile, B, H, W, C, N):
    """VAE Downsample (ldm/modules/diffusionmodules/model.py:72-76): F.pad(x,(0,1,0,1)) + conv3x3(stride 2, padding 0)."""
    g = _g(31)
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, 3, 3), g, 1.0 / math.sqrt(C * 9))
    x = a.float().reshape(B, H, W, C).permute(0, 3, 1, 2)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w.float(), None, stride=2, padding=0)
    Hout, Wout = ref.shape[2], ref.shape[3]
    assert (Hout, Wout) == (H // 2, W // 2)
    bias = torch.randn(N, generator=g)
    out32 = torch.full((B * Hout * Wout, N), float('nan'), device=DEV)
    K.igemm(a.to(DEV), K.pack_conv_weight(w.float().to(DEV)), N, B, H, W, Hout, Wout, 3, 2, 0, bias=bias.to(DEV),
            out_f32=out32, tile=tile, asym_pad=1)
    torch.cuda.synchronize()
    assert K.report(f'igemm asym-pad s2 tile{tile}', out32, _nhwc(ref) + bias[None], 2e-4) < 2e-4


def test_vae_small_kernels():
    g = _g(32)
    x = torch.randn(2, 4, 9, 7, generator=g)
    w = torch.randn(8, 4, generator=g)
    b = torch.randn(8, generator=g)
    ref = F.conv2d(x * 3.0, w[:, :, None, None], b)
    out = K.pointwise_nchw(x.to(DEV), w.to(DEV), b.to(DEV), in_scale=3.0)
    assert K.report('pointwise_nchw', out, ref, 1e-5) < 1e-5
    S = torch.randn(37, 4096, generator=g) * 40
    ref = torch.softmax(S * 0.044, dim=1)
    P = K.softmax_rows(S.to(DEV), 0.044)
    # fp16 output of values <= 1: half-ulp 2.4e-4 relative
    assert K.report('softmax_rows', P, ref, 5e-4) < 5e-4
    assert (P.float().sum(1) - 1).abs().max().item() < 2e-3


def test_groupnorm_large_map_and_range():
    """VAE-sized map (HW = 256*256, C = 128) and large-magnitude activations: the fixed-point statistics must neither
    lose precision nor wrap."""
    g = _g(33)
    B, HW, C = 1, 256 * 256, 128
    x = torch.randn(B, HW, C, generator=g) * 300.0 + 1000.0
    gamma = 1 + 0.1 * torch.randn(C, generator=g); beta = 0.1 * torch.randn(C, generator=g)
    ref = F.group_norm(x.permute(0, 2, 1).double(), 32, gamma.double(), beta.double(), eps=1e-6).permute(0, 2, 1).float()
    out = K.groupnorm(x.to(DEV), None, gamma.to(DEV), beta.to(DEV), 1e-6, 0, want=('f32',))['f32']
    # mean ~1000 with std 300: fp32 cancellation in (x - mean) * rstd costs ~1e-4 relative of |x|/std
    assert K.report('groupnorm 256x256x128 offset', out, ref, 2e-3) < 2e-3


def test_conv_in_out():
    g = _g(12)
    B, H, W = 2, 9, 12
    x = torch.randn(B, 4, H, W, generator=g)
    w = torch.randn(320, 4, 3, 3, generator=g) / 6
    b = torch.randn(320, generator=g) * 0.1
    ref = _nhwc(F.conv2d(x, w, b, padding=1)).reshape(B, H * W, 320)
    out = K.conv_in(x.to(DEV), w.to(DEV), b.to(DEV))
    assert K.report('conv_in', out, ref, 2e-5) < 2e-5
    h = torch.randn(B, 320, H, W, generator=g)
    w2 = torch.randn(4, 320, 3, 3, generator=g) / math.sqrt(2880)
    b2 = torch.randn(4, generator=g) * 0.1
    ref2 = F.conv2d(h, w2, b2, padding=1)
    out2 = K.conv_out(_nhwc(h).reshape(B, H * W, 320).to(DEV), w2.to(DEV), b2.to(DEV), B, H, W)
    assert K.report('conv_out', out2, ref2, 2e-5) < 2e-5


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 9, 12, 320, 4), (1, 16, 16, 128, 3), (2, 8, 8, 512, 8), (1, 5, 20, 64, 4), (2, 64, 64, 320, 4)])
def test_conv_out_4_pixels_per_wave(B, H, W, Cin, Cout, monkeypatch):
    """conv_out4_kernel (one wave per 4 pixels of a row; UNet `out` head openaimodel.py:533-537, first-stage conv_out
    model.py:553-566, encoder conv_out) vs F.conv2d, and against the one-wave-per-pixel kernel: same terms in the same order
    per accumulator, but the compiler contracts the FMAs differently -- a few fp32 ulp, on the last op of the path."""
    g = _g(31)
    h = torch.randn(B, Cin, H, W, generator=g)
    w2 = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b2 = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(h, w2, b2, padding=1)
    hd = _nhwc(h).reshape(B, H * W, Cin).to(DEV)
    monkeypatch.setenv('SDMI_CONV_OUT4', '1')
    out4 = K.conv_out(hd, w2.to(DEV), b2.to(DEV), B, H, W).clone()
    monkeypatch.setenv('SDMI_CONV_OUT4', '0')
    out1 = K.conv_out(hd, w2.to(DEV), b2.to(DEV), B, H, W).clone()
    torch.cuda.synchronize()
    assert K.report(f'conv_out4 {B}x{Cin}x{H}x{W}->{Cout}', out4, ref, 2e-5) < 2e-5
    assert float((out4 - out1).abs().max()) <= 4e-6


@pytest.mark.parametrize('B,N,K_', [(2, 1280, 320), (2, 1280, 1280), (2, 20160, 1280), (5, 77, 64), (8, 640, 1280)])
@pytest.mark.parametrize('silu', [0, 1])
def test_small_linear_lds_staged(B, N, K_, silu, monkeypatch):
    """small_linear_lds_kernel (activated input rows staged once per block, all weight quads of a lane in flight; time_embed and
    the 22 emb_layers, openaimodel.py:506-511,218-224) vs F.linear, and bit for bit against the one-load-at-a-time kernel."""
    g = _g(32)
    w = torch.randn(N, K_, generator=g) / math.sqrt(K_)
    b = torch.randn(N, generator=g) * 0.1
    x = torch.randn(B, K_, generator=g) * 2
    ref = F.linear(F.silu(x) if silu else x, w, b)
    monkeypatch.setenv('SDMI_SMALL_LDS', '1')
    o1 = K.small_linear(x.to(DEV), w.to(DEV), b.to(DEV), silu).clone()
    monkeypatch.setenv('SDMI_SMALL_LDS', '0')
    o0 = K.small_linear(x.to(DEV), w.to(DEV), b.to(DEV), silu).clone()
    torch.cuda.synchronize()
    assert K.report(f'small_linear lds B{B} N{N} K{K_} silu{silu}', o1, ref, 3e-5) < 3e-5
    assert torch.equal(o1, o0)


@pytest.mark.parametrize('mode', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('cfg', [0, 1])
def test_sampler_step_bit_exact(mode, cfg):
    """CFG combine + PLMS/DDIM update (plms.py:178-236): bit-identical to the reference's fp32 torch expression."""
    g = _g(13 + mode)
    n = (2, 4, 16, 16)
    x = torch.randn(n, generator=g)
    eps = torch.randn((4,) + n[1:], generator=g) if cfg else torch.randn(n, generator=g)
    old = [torch.randn(n, generator=g) for _ in range(3)]
    a_t, a_prev, sigma = 0.4321, 0.5678, (0.1 if mode == 0 else 0.0)
    s1m = float(torch.tensor(1.0 - a_t).sqrt())
    noise = torch.randn(n, generator=g) if sigma else None
    scale = 7.5
    if cfg:
        eu, ec = eps.chunk(2)
        e_t = eu + scale * (ec - eu)
    else:
        e_t = eps
    if mode == 0: ep = e_t
    elif mode == 1: ep = (3 * e_t - old[0]) / 2
    elif mode == 2: ep = (23 * e_t - 16 * old[0] + 5 * old[1]) / 12
    elif mode == 3: ep = (55 * e_t - 59 * old[0] + 37 * old[1] - 9 * old[2]) / 24
    else: ep = (old[0] + e_t) / 2
    b = n[0]
    at = torch.full((b, 1, 1, 1), a_t); ap = torch.full((b, 1, 1, 1), a_prev)
    sg = torch.full((b, 1, 1, 1), sigma); sm = torch.full((b, 1, 1, 1), s1m)
    pred = (x - sm * ep) / at.sqrt()
    dir_xt = (1. - ap - sg ** 2).sqrt() * ep
    xp = ap.sqrt() * pred + dir_xt + (sg * noise if noise is not None else 0.)
    e_o, x_o, p_o = K.sampler_step(eps.to(DEV), cfg, scale, x.to(DEV), mode, [o.to(DEV) for o in old], a_t, a_prev, sigma,
                                   s1m, None if noise is None else noise.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(e_o.cpu(), e_t), 'post-CFG eps not bit-exact'
    assert torch.equal(p_o.cpu(), pred), 'pred_x0 not bit-exact'
    assert torch.equal(x_o.cpu(), xp), 'x_prev not bit-exact'


# ---- GroupNorm + SiLU folded into the halo staging of the 3x3 convolution (csrc/conv3halo.hip, conv3halo_gn_kernel) ---------------
GN_FOLD_CASES = [
    # name, B, H, W, c0, c1, N, splitk, tile  (at least 8 channels per group: c0 + c1 >= 256)
    ('l0_w64_t14', 2, 64, 64, 320, 0, 64, 1, 14),
    ('l0_w64_t15', 1, 64, 64, 256, 0, 128, 2, 15),
    ('l0_w64_t16_cat', 1, 64, 64, 192, 128, 64, 1, 16),       # two fp32 sources with different row pitches
    ('l1_w32_t14_cat', 2, 32, 32, 640, 320, 128, 3, 14),
    ('l1_w32_t17', 2, 32, 32, 320, 0, 128, 1, 17),
    ('l2_w16_t14', 2, 16, 16, 1280, 0, 128, 5, 14),           # one whole 16x16 image per 256-row tile
    ('l2_w16_t16', 2, 16, 16, 640, 0, 64, 2, 16),
    ('l3_w8_t16', 2, 8, 8, 1280, 0, 128, 10, 16),             # two whole 8x8 images per 128-row tile
    ('l3_w8_t14_b4', 4, 8, 8, 640, 640, 64, 4, 14),           # four whole images per 256-row tile
    ('auto', 2, 32, 32, 640, 0, 640, 0, -1),                  # tile and split chosen by the launcher
    # a workgroup's chunk range crossing from the first fp32 source into the second, small channel counts (the TINY / SMALL40
    # configurations' output blocks), every tile at W = 64
    ('l0_w64_t16', 1, 64, 64, 320, 0, 64, 1, 16),
    ('l0_w64_t14_cat', 1, 64, 64, 192, 128, 64, 1, 14),
    ('l0_w64_t17_cat', 1, 64, 64, 192, 128, 128, 1, 17),
    ('l1_w32_t16_cat', 1, 32, 32, 192, 128, 64, 1, 16),
    ('l1_w32_t14_cat_cross', 2, 32, 32, 640, 320, 128, 1, 14),
    ('tiny_w16_t14_cat', 2, 16, 16, 128, 128, 128, 2, 14),
    ('tiny_w16_t14_cat_k1', 2, 16, 16, 128, 128, 128, 1, 14),
    ('tiny_w8_t16_cat', 2, 8, 8, 128, 128, 64, 1, 16),
    ('tiny_w8_t16_cat_k2', 2, 8, 8, 128, 128, 64, 2, 16),
]


def _gn_fold_inputs(B, H, W, c0, c1, N, seed):
    g = _g(seed)
    C = c0 + c1
    x0 = torch.randn(B, H, W, c0, generator=g) * 1.3 + 0.2
    x1 = torch.randn(B, H, W, c1, generator=g) * 0.8 - 0.1 if c1 else None
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    w = torch.randn(N, C, 3, 3, generator=g) / math.sqrt(9 * C)
    bias = torch.randn(N, generator=g) * 0.1
    rowvec = torch.randn(B, N, generator=g)
    resid = torch.randn(B * H * W, N, generator=g)
    return x0, x1, gamma, beta, w, bias, rowvec, resid


@pytest.mark.experiments
@pytest.mark.parametrize('case', GN_FOLD_CASES, ids=[c[0] for c in GN_FOLD_CASES])
def test_conv3_gn_fold(case):
    """ResBlock in_layers / out_layers: conv3x3(SiLU(GroupNorm32(cat(x0, x1)))) + bias + emb + residual
    (openaimodel.py:201-204,225-231,263-275) as ONE launch that normalises its input while staging it: against the fp32 torch
    ops, and BIT FOR BIT against the two-launch path it replaces (GroupNorm-apply kernel -> LDS-DMA halo conv, same tile)."""
    name, B, H, W, c0, c1, N, splitk, tile = case
    x0, x1, gamma, beta, w, bias, rowvec, resid = _gn_fold_inputs(B, H, W, c0, c1, N, B * H + c0 + N)
    C = c0 + c1
    x = x0 if x1 is None else torch.cat([x0, x1], dim=3)
    xn = F.silu(F.group_norm(x.permute(0, 3, 1, 2), 32, gamma, beta, 1e-5))
    # the kernel rounds the normalised activation and the weights to fp16 once (MFMA operands), accumulates in fp32
    ref = F.conv2d(xn.half().float(), w.half().float(), None, padding=1)
    ref = _nhwc(ref) + bias[None] + rowvec.repeat_interleave(H * W, dim=0) + resid
    d = lambda t: None if t is None else t.to(DEV)
    out, rhi, rlo = K.conv3gn(d(x0), d(x1), d(gamma), d(beta), 1e-5, d(w), bias=d(bias), rowvec=d(rowvec), residual=d(resid),
                              splitk=splitk, tile=tile, want_raw=True)
    torch.cuda.synchronize()
    # vs the fp16-operand reference: fp32 GN rounding flips of fp16 operands (<= 1 fp16 ulp on a handful of A elements) and
    # accumulation order
    assert K.report(f'conv3 gn-fold {name}', out, ref, 3e-3) < 3e-3
    # the raw split-fp16 copy (operand of the ResBlock's 1x1 skip conv): exactly the cast kernel's values, every pixel once
    xc = x.reshape(B * H * W, C).to(DEV)
    hi, lo = K.cast_f16(xc, want_lo=True)
    assert torch.equal(rhi, hi) and torch.equal(rlo, lo)
    if tile >= 0:
        # the path it replaces, same tile and split: identical fp16 operand -> identical MFMA sequence -> identical bits
        gn = K.groupnorm(d(x0).reshape(B, H * W, c0), None if x1 is None else d(x1).reshape(B, H * W, c1), d(gamma), d(beta), 1e-5, 1,
                         want=('f16',))
        out2 = torch.full((B * H * W, N), float('nan'), device=DEV)
        K.igemm(gn['f16'].reshape(B * H * W, C), K.pack_conv_weight(d(w)), N, B, H, W, H, W, ksize=3, bias=d(bias), rowvec=d(rowvec),
                residual=d(resid), out_f32=out2, splitk=splitk, tile=tile, fused_splitk=False)
        torch.cuda.synchronize()
        assert torch.equal(out, out2), float((out - out2).abs().max())


@pytest.mark.parametrize('ptile,ctile', [(-1, -1), (5, 5), (4, 8), (1, 0), (10, 3)])
@pytest.mark.parametrize('mode', ['heads', 'geglu', 'plain'])
def test_layernorm_folded_into_consumer(mode, ptile, ctile):
    """x = residual + a @ Wo^T + b (the token stream, attention.py:212-214), then Linear(LayerNorm(x)) -- q|k|v projection,
    GEGLU or a plain Linear -- WITHOUT a LayerNorm launch: the producing GEMM stores fp16(gamma * x) and per-row {sum, sumsq}
    partials, the consuming GEMM turns its accumulators into rstd * (acc - mean * cs) + d.  Against fp32 torch
    (F.layer_norm -> F.linear) and against the two-launch path (producer -> layernorm kernel -> GEMM)."""
    g = _g(77 + ptile + ctile)
    B, ntok, Kp, C = 2, 1024, 256, 320                      # M = 2048 rows of C channels
    M = B * ntok
    a = (torch.randn(M, Kp, generator=g) * 0.7).half()
    wo = (torch.randn(C, Kp, generator=g) / math.sqrt(Kp)).half()
    bo = torch.randn(C, generator=g) * 0.1
    resid = torch.randn(M, C, generator=g) * 1.5 + 0.3       # (a row mean that is not zero)
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    x = resid + a.float() @ wo.float().t() + bo
    xn = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    d = lambda t: t.to(DEV)
    x_dev = torch.empty(M, C, device=DEV); a16 = torch.empty(M, C, dtype=torch.float16, device=DEV)
    part = torch.full((C // 32, M, 2), float('nan'), device=DEV)
    K.igemm(d(a), d(wo), C, B, ntok, 1, ntok, 1, bias=d(bo), residual=d(resid), out_f32=x_dev, out_f16=a16, tile=ptile,
            f16_scale=d(gamma), lnp_out=part)
    torch.cuda.synchronize()
    assert K.report('ln-fold producer x', x_dev, x, 2e-3) < 2e-3
    # the partials: sums over 32-column blocks of the stored fp32 values
    xs = x_dev.cpu().double().reshape(M, C // 32, 32)
    assert torch.allclose(part[..., 0].cpu().double().t(), xs.sum(-1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(part[..., 1].cpu().double().t(), (xs * xs).sum(-1), rtol=1e-5, atol=1e-3)
    assert torch.equal(a16.cpu(), (x_dev.cpu() * gamma).half())
    # reference operand of the two-launch path
    ln16 = K.layernorm(x_dev, d(gamma), d(beta))
    if mode == 'plain':
        N = 640
        w = (torch.randn(N, C, generator=g) / math.sqrt(C)).half(); b = torch.randn(N, generator=g) * 0.1
        ref = xn @ w.float().t() + b
        cs, dn = K.ln_fold_prep(d(w), C, d(gamma), d(beta), d(b))
        out = torch.empty(M, N, device=DEV); out2 = torch.empty(M, N, device=DEV)
        K.igemm(a16, d(w), N, B, ntok, 1, ntok, 1, out_f32=out, tile=ctile, lnf=(part, 1e-5, cs, dn))
        K.igemm(ln16, d(w), N, B, ntok, 1, ntok, 1, bias=d(b), out_f32=out2, tile=ctile)
        torch.cuda.synchronize()
        e1 = K.report(f'ln-fold plain folded   p{ptile} c{ctile}', out, ref, 6e-3)
        e2 = K.report(f'ln-fold plain two-step p{ptile} c{ctile}', out2, ref, 6e-3)
    elif mode == 'geglu':
        N = 8 * C
        w = torch.randn(N, C, generator=g) / math.sqrt(C); b = torch.randn(N, generator=g) * 0.1
        y = xn @ w.half().float().t() + b
        ref = y[:, :N // 2] * F.gelu(y[:, N // 2:])
        wp, bp = K.pack_geglu(d(w), d(b))
        cs, dn = K.ln_fold_prep(wp, C, d(gamma), d(beta), bp)
        out = torch.empty(M, N // 2, dtype=torch.float16, device=DEV); out2 = torch.empty_like(out)
        ct = ctile if ctile in (-1, 0, 3, 8) else 0
        K.igemm(a16, wp, N, B, ntok, 1, ntok, 1, out_f16=out, mode=1, tile=ct, lnf=(part, 1e-5, cs, dn))
        K.igemm(ln16, wp, N, B, ntok, 1, ntok, 1, bias=bp, out_f16=out2, mode=1, tile=ct)
        torch.cuda.synchronize()
        e1 = K.report(f'ln-fold geglu folded   p{ptile} c{ct}', out.float(), ref, 8e-3)
        e2 = K.report(f'ln-fold geglu two-step p{ptile} c{ct}', out2.float(), ref, 8e-3)
    else:
        heads, dh = 8, C // 8
        w = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).half()
        ref = xn @ w.float().t()
        cs, dn = K.ln_fold_prep(d(w), C, d(gamma), d(beta))
        outs = []
        for src, extra in ((a16, dict(lnf=(part, 1e-5, cs, dn))), (ln16, {})):
            q = torch.empty(B * heads, ntok, dh, dtype=torch.float16, device=DEV); k = torch.empty_like(q)
            vt = torch.empty(B * heads, dh, ntok, dtype=torch.float16, device=DEV)
            K.igemm(src, d(w), 3 * C, B, ntok, 1, ntok, 1, mode=2, tile=ctile,
                    heads=dict(segs=[(q, 0), (k, 0), (vt, 1)], heads=heads, dh=dh, ntok=ntok, ntok_pad=ntok, segC=C), **extra)
            torch.cuda.synchronize()
            qq = q.float().reshape(B, heads, ntok, dh).permute(0, 2, 1, 3).reshape(M, C)
            kk = k.float().reshape(B, heads, ntok, dh).permute(0, 2, 1, 3).reshape(M, C)
            vv = vt.float().reshape(B, heads, dh, ntok).permute(0, 3, 1, 2).reshape(M, C)
            outs.append(torch.cat([qq, kk, vv], dim=1))
        e1 = K.report(f'ln-fold heads folded   p{ptile} c{ctile}', outs[0], ref, 6e-3)
        e2 = K.report(f'ln-fold heads two-step p{ptile} c{ctile}', outs[1], ref, 6e-3)
    # the fold must not cost accuracy: the two-launch path's error level (both round one fp16 operand per element; the maxima of
    # 1-3 M outputs scatter by +-20 %)
    assert e1 < 8e-3 and e1 <= 1.5 * e2 + 5e-4, (e1, e2)


FIVE_WAVE_CASES = [c for c in CONV_CASES if not c[9]]        # (the five-wave tile has no upsampling gather)


@pytest.mark.experiments
@pytest.mark.parametrize('case', FIVE_WAVE_CASES, ids=[c[0] for c in FIVE_WAVE_CASES])
@pytest.mark.parametrize('splitk', [1, 3])
def test_igemm_five_wave_tile(case, splitk):
    """tile 22 (csrc/igemm5.hip): 64 x 160, five waves side by side, operands dealt in 8-row octets over the (A | W) rows -- the
    conv / linear cases of test_igemm_conv incl. masked taps, stride 2, concatenated sources, M and N tails, with and without
    split-K (separate reduce)."""
    name, B, Hin, Win, c0, c1, N, ksize, stride, up = case
    g = _g(hash(name) % 1000 + 5)
    Cin = c0 + c1
    big = _rand16((B * Hin * Win, Cin), g)
    a0 = big[:, :c0]
    a1 = big[:, c0:] if c1 else None
    w = _rand16((N, Cin, ksize, ksize), g, 1.0 / math.sqrt(Cin * ksize * ksize))
    ref = _conv_ref(a0, a1, w, B, Hin, Win, ksize, stride, up)
    Hout, Wout = ref.shape[2], ref.shape[3]
    M = B * Hout * Wout
    bias = torch.randn(N, generator=g)
    rowvec = torch.randn(B, N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ref2 = _nhwc(ref) + bias[None] + rowvec.repeat_interleave(Hout * Wout, dim=0) + resid
    wp = K.pack_conv_weight(w.float().to(DEV))
    nkt = (ksize * ksize * Cin) // 64
    if splitk > 1 and (N % 4 or nkt < splitk):
        pytest.skip('split-K needs N % 4 == 0 and enough k-tiles')
    out32 = torch.full((M, N), float('nan'), device=DEV)
    out16 = torch.full((M, N), float('nan'), device=DEV, dtype=torch.float16)
    big_d = big.to(DEV)
    K.igemm(big_d[:, :c0], wp, N, B, Hin, Win, Hout, Wout, ksize, stride, up, a1=big_d[:, c0:] if c1 else None,
            bias=bias.to(DEV), rowvec=rowvec.to(DEV), residual=resid.to(DEV), out_f32=out32, out_f16=out16,
            tile=22, splitk=splitk, fused_splitk=False)
    torch.cuda.synchronize()
    assert K.report(f'igemm5 {name} split{splitk} f32', out32, ref2, 2e-4) < 2e-4
    assert K.report(f'igemm5 {name} split{splitk} f16', out16, ref2, 6e-3) < 6e-3


@pytest.mark.experiments
@pytest.mark.parametrize('B,H,W,C,N,ksize', [(2, 64, 64, 320, 320, 3), (2, 64, 64, 320, 320, 1), (2, 32, 32, 640, 640, 3), (2, 64, 64, 320, 960, 1)])
def test_igemm_five_wave_tile_sd_shapes(B, H, W, C, N, ksize):
    """the shapes tile 22 is for (64x64 / 32x32 levels of SD v1), with the GroupNorm statistics of the output from the epilogue:
    against tile 5 (64 x 64) on the same operands -- same products, per-k-tile accumulation order, so the outputs agree to the
    last bits of an fp32 sum -- and the statistics against torch."""
    g = _g(C + N + ksize)
    a = _rand16((B * H * W, C), g)
    w = _rand16((N, C, ksize, ksize), g, 1.0 / math.sqrt(C * ksize * ksize))
    bias = torch.randn(N, generator=g)
    resid = torch.randn(B * H * W, N, generator=g)
    wp = K.pack_conv_weight(w.float().to(DEV))
    outs, accs = [], []
    for tile in (22, 5):
        out = torch.full((B * H * W, N), float('nan'), device=DEV)
        acc = torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
        gn = [(acc, N // 32, 0)] if N // 32 >= 2 else None
        K.igemm(a.to(DEV), wp, N, B, H, W, H, W, ksize, bias=bias.to(DEV), residual=resid.to(DEV), out_f32=out, tile=tile, gn=gn)
        torch.cuda.synchronize()
        outs.append(out); accs.append(acc)
    assert K.report(f'igemm5 sd M{B * H * W} N{N} K{C * ksize * ksize} vs tile 5', outs[0], outs[1], 1e-5) < 1e-5
    s0, q0 = K.gn_acc_sums(accs[0])
    xs = outs[0].cpu().double().reshape(B, H * W, 32, N // 32)
    assert torch.allclose(s0, xs.sum(dim=(1, 3)), rtol=1e-6, atol=1e-3)
    assert torch.allclose(q0, (xs * xs).sum(dim=(1, 3)), rtol=1e-6, atol=1e-2)


@pytest.mark.experiments
@pytest.mark.parametrize('d,heads,nq,nkv,B', [(40, 8, 4096, 77, 2), (80, 8, 1024, 77, 2), (160, 8, 256, 77, 2), (160, 8, 64, 77, 2),
                                              (40, 8, 100, 77, 1), (80, 8, 64, 128, 1), (160, 8, 33, 5, 3), (40, 8, 256, 96, 2)])
@pytest.mark.parametrize('fold', [False, True])
def test_attention_ctx_fused_q(d, heads, nq, nkv, B, fold, monkeypatch):
    """Cross-attention with the to_q projection inside the kernel (csrc/attn_ctx.hip; attention.py:161,170-193):
    out = softmax((LN(t) Wq^T) K^T d^-0.5) V over the cached context keys, against fp32 torch and against the two-launch path
    (to_q GEMM with the per-head scatter -> flash attention); with and without the LayerNorm fold (x = fp16(gamma t) + row partials)."""
    g = _g(d + nq + nkv + (7 if fold else 0))
    C = heads * d
    M = B * nq
    t = torch.randn(M, C, generator=g) * 1.2 + 0.2
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).half()
    nkv_pad = (nkv + 7) // 8 * 8
    k = (torch.randn(B * heads, nkv, d, generator=g) * 0.8).half()
    v = (torch.randn(B * heads, nkv, d, generator=g)).half()
    vt = torch.zeros(B * heads, d, nkv_pad, dtype=torch.float16)
    vt[:, :, :nkv] = v.transpose(1, 2)
    scale = d ** -0.5
    xn = F.layer_norm(t, (C,), gamma, beta, 1e-5)
    q = (xn @ wq.float().t()).reshape(B, nq, heads, d).permute(0, 2, 1, 3).reshape(B * heads, nq, d)
    att = torch.softmax(q @ k.float().transpose(1, 2) * scale, dim=-1) @ v.float()
    ref = att.reshape(B, heads, nq, d).permute(0, 2, 1, 3).reshape(B, nq, C)
    dv = lambda z: z.to(DEV)
    ln16 = K.layernorm(dv(t), dv(gamma), dv(beta))
    if fold:
        if C > 640:
            pytest.skip('the fold is taken for C <= 640 (20 partials)')
        # the producer's side, emulated: fp16(gamma * t) and {sum, sumsq} per 32-column block
        x16 = (t * gamma).half()
        tb = t.double().reshape(M, C // 32, 32)
        part = torch.stack([tb.sum(-1), (tb * tb).sum(-1)], dim=-1).permute(1, 0, 2).float().contiguous()
        cs, dn = K.ln_fold_prep(dv(wq), C, dv(gamma), dv(beta))
        out = K.attention_ctx(dv(x16), dv(wq), dv(k), dv(vt), heads, nkv, scale, lnf=(dv(part), 1e-5, cs, dn))
    else:
        out = K.attention_ctx(ln16, dv(wq), dv(k), dv(vt), heads, nkv, scale)
    # the two-launch path on the same operands
    qd = torch.empty(B * heads, nq, d, dtype=torch.float16, device=DEV)
    K.igemm(ln16, dv(wq), C, B, nq, 1, nq, 1, mode=2,
            heads=dict(segs=[(qd, 0)], heads=heads, dh=d, ntok=nq, ntok_pad=(nq + 7) // 8 * 8, segC=C))
    out2 = K.attention(qd, dv(k), dv(vt), heads, nkv, scale)
    torch.cuda.synchronize()
    e1 = K.report(f'attn_ctx fused d{d} nq{nq} nkv{nkv} B{B} fold{int(fold)}', out, ref, 4e-3)
    e2 = K.report(f'attn_ctx two-launch d{d} nq{nq} nkv{nkv} B{B}', out2, ref, 4e-3)
    assert not torch.isnan(out).any()
    assert e1 < 4e-3 and e1 <= 1.5 * e2 + 3e-4, (e1, e2)


@pytest.mark.experiments
@pytest.mark.parametrize('d,heads,nq,nkv', [(40, 8, 4096, 4096), (40, 8, 4096, 4000), (80, 8, 1024, 1024), (64, 4, 512, 77), (40, 8, 2304, 2304),
                                            (128, 2, 300, 130), (32, 4, 256, 64)])
def test_attention_pingpong_is_bit_identical(d, heads, nq, nkv, monkeypatch):
    """attn_pp_kernel (SDMI_ATTN_PP=1): the halves of an 8-wave workgroup alternate their matrix and VALU blocks instead of running
    in lock-step -- per wave the same instructions on the same values as attn_dma_kernel, so the output must not change by one bit
    (and both are within the usual tolerance of fp32 torch)."""
    monkeypatch.setenv('SDMI_ATTN_KVS', '0')     # (the 8-wave kernel is the one it mirrors; round 6's default at 4096 keys is the key-split kernel)
    g = _g(d + nq + nkv)
    B = 2
    q = (torch.randn(B * heads, nq, d, generator=g)).half()
    k = (torch.randn(B * heads, nkv, d, generator=g)).half()
    v = (torch.randn(B * heads, nkv, d, generator=g)).half()
    nkv_pad = (nkv + 7) // 8 * 8
    vt = torch.zeros(B * heads, d, nkv_pad, dtype=torch.float16)
    vt[:, :, :nkv] = v.transpose(1, 2)
    scale = d ** -0.5
    ref = (torch.softmax(q.float() @ k.float().transpose(1, 2) * scale, dim=-1) @ v.float())
    ref = ref.reshape(B, heads, nq, d).permute(0, 2, 1, 3).reshape(B, nq, heads * d)
    monkeypatch.setenv('SDMI_ATTN_NW', '8')
    monkeypatch.setenv('SDMI_ATTN_PP', '0')
    o0 = K.attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads, nkv, scale).clone()
    monkeypatch.setenv('SDMI_ATTN_PP', '1')
    o1 = K.attention(q.to(DEV), k.to(DEV), vt.to(DEV), heads, nkv, scale).clone()
    torch.cuda.synchronize()
    assert K.report(f'attention ping-pong d{d} nq{nq} nkv{nkv}', o1, ref, 3e-3) < 3e-3
    assert torch.equal(o0, o1), float((o0.float() - o1.float()).abs().max())
