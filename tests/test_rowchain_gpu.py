"""Row-strip chain kernels (csrc/rowchain.hip) against the launches they replace and against fp32 torch.

ff_tail: GEGLU -> FF-out -> proj_out of a SpatialTransformer (ldm/modules/attention.py:58-64,214,258-261) as ONE launch.  The fused
kernel performs the same MFMA accumulations in the same order as the three unsplit launches, so its outputs are compared bit for
bit; the GroupNorm statistics are sums of differently grouped fp32 partials (fixed-point accumulated) and are compared to 1e-6."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


import kernels as K  # noqa: E402  (tests/ is on sys.path via conftest)

DEV = 'cuda'


def _g(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def _ff_tail_case(B, ntok, seed):
    g = _g(seed)
    C_ = 320
    M = B * ntok
    d = lambda t: t.to(DEV)
    ao = (torch.randn(M, C_, generator=g) * 0.7).half()
    wo2 = (torch.randn(C_, C_, generator=g) / math.sqrt(C_)).half()
    bo2 = torch.randn(C_, generator=g) * 0.1
    t_prev = torch.randn(M, C_, generator=g) * 1.5 + 0.3
    gamma = 1 + 0.2 * torch.randn(C_, generator=g)
    beta = 0.1 * torch.randn(C_, generator=g)
    wgg = torch.randn(8 * C_, C_, generator=g) / math.sqrt(C_)
    bgg = torch.randn(8 * C_, generator=g) * 0.1
    wff2 = (torch.randn(C_, 4 * C_, generator=g) / math.sqrt(4 * C_)).half()
    bff2 = torch.randn(C_, generator=g) * 0.1
    wpo = torch.randn(C_, C_, generator=g) / math.sqrt(C_)
    bpo = torch.randn(C_, generator=g) * 0.1
    x_in = torch.randn(M, C_, generator=g) * 2.0
    # producer of the token stream: t = t_prev + ao Wo^T + b, with the LayerNorm-fold side outputs (attention.py:213)
    t = torch.empty(M, C_, device=DEV); ln16 = torch.empty(M, C_, dtype=torch.float16, device=DEV)
    part = torch.full((C_ // 32, M, 2), float('nan'), device=DEV)
    K.igemm(d(ao), d(wo2), C_, B, ntok, 1, ntok, 1, bias=d(bo2), residual=d(t_prev), out_f32=t, out_f16=ln16, f16_scale=d(gamma),
            lnp_out=part)
    wp, bp = K.pack_geglu(d(wgg), d(bgg))
    cs, dn = K.ln_fold_prep(wp, C_, d(gamma), d(beta), bp)
    csd = torch.cat([cs.view(4, 2 * C_), dn.view(4, 2 * C_)], dim=1).contiguous()
    wpo3 = K.pack_split3(d(wpo))
    torch.cuda.synchronize()
    return dict(B=B, ntok=ntok, M=M, C=C_, t=t, ln16=ln16, part=part, wp=wp, cs=cs, dn=dn, csd=csd, wff2=d(wff2), bff2=d(bff2), wpo3=wpo3,
                bpo=d(bpo), x_in=d(x_in), gamma=gamma, beta=beta, wgg=wgg, bgg=bgg, wpo=wpo)


def _three_launches(c, gn=None, want_copy=True):
    M, C_, B, ntok = c['M'], c['C'], c['B'], c['ntok']
    gg = torch.empty(M, 4 * C_, dtype=torch.float16, device=DEV)
    K.igemm(c['ln16'], c['wp'], 8 * C_, B, ntok, 1, ntok, 1, out_f16=gg, mode=1, lnf=(c['part'], 1e-5, c['cs'], c['dn']))
    hi = torch.empty(M, C_, dtype=torch.float16, device=DEV); lo = torch.empty_like(hi)
    K.igemm(gg, c['wff2'], C_, B, ntok, 1, ntok, 1, bias=c['bff2'], residual=c['t'], out_f16=hi, out_lo=lo)
    out = torch.full((M, C_), float('nan'), device=DEV)
    copy = torch.empty(M, C_, dtype=torch.float16, device=DEV) if want_copy else None
    K.igemm(hi, c['wpo3'], C_, B, ntok, 1, ntok, 1, a1=lo, bias=c['bpo'], residual=c['x_in'], out_f32=out, out_f16=copy, split16=True, gn=gn)
    return out, copy, gg, hi, lo


@pytest.mark.parametrize('B,ntok', [(2, 4096), (1, 64), (3, 128), (2, 9216)])
def test_ff_tail_is_bit_identical_to_the_three_launches(B, ntok):
    c = _ff_tail_case(B, ntok, 1234 + ntok)
    M, C_ = c['M'], c['C']
    mk = lambda: torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
    # statistics targets as the UNet attaches them: the next ResBlock's GroupNorm over this tensor alone (10 channels per group)
    # and a skip-concat GroupNorm where it is channels [640, 960) of 960 (30 per group: groups straddle the 32-column tiles)
    acc_a, acc_b, ref_a, ref_b = mk(), mk(), mk(), mk()
    out_ref, copy_ref, gg, hi, lo = _three_launches(c, gn=[(ref_a, 10, 0), (ref_b, 30, 640)])
    out = torch.full((M, C_), float('nan'), device=DEV)
    copy = torch.full((M, C_), float('nan'), dtype=torch.float16, device=DEV)
    K.ff_tail(c['ln16'], c['part'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['t'], c['wpo3'], c['bpo'], c['x_in'], out, B, ntok,
              out_f16=copy, gn=[(acc_a, 10, 0), (acc_b, 30, 640)])
    torch.cuda.synchronize()
    # against fp32 torch (sanity of the math; both paths round the same operands to fp16)
    tt = c['t'].cpu()
    y = F.layer_norm(tt, (C_,), c['gamma'], c['beta'], 1e-5) @ c['wgg'].half().float().t() + c['bgg']
    ff = (y[:, :4 * C_] * F.gelu(y[:, 4 * C_:])) @ c['wff2'].cpu().float().t() + c['bff2'].cpu()
    ref = c['x_in'].cpu() + (tt + ff) @ c['wpo'].t() + c['bpo'].cpu()
    e_ref = K.report(f'ff_tail three launches B{B} n{ntok}', out_ref, ref, 3e-2)
    e_new = K.report(f'ff_tail one launch     B{B} n{ntok}', out, ref, 3e-2)
    assert e_new < 3e-2 and e_new <= e_ref * 1.2 + 1e-4
    dmax = float((out - out_ref).abs().max())
    print(f'[ff_tail vs launches] max diff {dmax:.3e}  equal={torch.equal(out, out_ref)} copy_equal={torch.equal(copy, copy_ref)}', flush=True)
    assert torch.equal(out, out_ref), dmax
    assert torch.equal(copy, copy_ref)
    for got, want, cpg, cbase in ((acc_a, ref_a, 10, 0), (acc_b, ref_b, 30, 640)):
        s, ss = K.gn_acc_sums(got)
        s0, ss0 = K.gn_acc_sums(want)
        assert torch.allclose(s, s0, rtol=1e-6, atol=1e-3) and torch.allclose(ss, ss0, rtol=1e-6, atol=1e-3)
        # ... and against float64 sums of the stored output
        o = out.cpu().double().reshape(B, ntok, C_)
        gid = (cbase + torch.arange(C_)) // cpg
        es = torch.zeros(B, 32, dtype=torch.float64); ess = torch.zeros(B, 32, dtype=torch.float64)
        es.index_add_(1, gid, o.sum(1)); ess.index_add_(1, gid, (o * o).sum(1))
        assert torch.allclose(s, es, rtol=1e-5, atol=1e-2) and torch.allclose(ss, ess, rtol=1e-5, atol=1e-1)


def test_ff_tail_without_side_outputs():
    c = _ff_tail_case(2, 256, 99)
    out_ref, _, _, _, _ = _three_launches(c, gn=None, want_copy=False)
    out = torch.full((c['M'], c['C']), float('nan'), device=DEV)
    K.ff_tail(c['ln16'], c['part'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['t'], c['wpo3'], c['bpo'], c['x_in'], out, 2, 256)
    torch.cuda.synchronize()
    assert torch.equal(out, out_ref), float((out - out_ref).abs().max())


def test_ff_tail_repeats_bit_identically_next_to_other_work():
    """50 launches interleaved with a cache-thrashing kernel: the counted LDS-DMA waits must hold with cold weights too."""
    c = _ff_tail_case(2, 4096, 7)
    out0 = torch.full((c['M'], c['C']), float('nan'), device=DEV)
    K.ff_tail(c['ln16'], c['part'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['t'], c['wpo3'], c['bpo'], c['x_in'], out0, 2, 4096)
    junk = torch.empty(96 << 20, device=DEV)          # 384 MB > the 256 MB Infinity Cache
    for i in range(50):
        if i % 2:
            junk.fill_(float(i))
        out = torch.full((c['M'], c['C']), float('nan'), device=DEV)
        K.ff_tail(c['ln16'], c['part'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['t'], c['wpo3'], c['bpo'], c['x_in'], out, 2, 4096)
        assert torch.equal(out, out0), (i, float((out - out0).abs().max()))
