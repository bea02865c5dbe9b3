"""Row-strip chain kernels (csrc/rowchain.hip) against the launches they replace and against fp32 torch.

ff_tail: GEGLU -> FF-out -> proj_out of a SpatialTransformer (ldm/modules/attention.py:58-64,214,258-261) as ONE launch.  The fused
kernel performs the same MFMA accumulations in the same order as the three unsplit launches, so its outputs are compared bit for
bit; the GroupNorm statistics are sums of differently grouped fp32 partials (fixed-point accumulated) and are compared to 1e-6.

st_head: GroupNorm -> proj_in -> (norm1) q | k | v (attention.py:254-256, 212, 170-176) as ONE launch, bit for bit against the
GroupNorm-apply + split-fp16 GEMM + q|k|v GEMM launches, and against fp32 torch."""
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
                bpo=d(bpo), x_in=d(x_in), gamma=gamma, beta=beta, wgg=wgg, bgg=bgg, wpo=wpo, ao=d(ao), wo2=d(wo2), bo2=d(bo2), t_prev=d(t_prev),
                dgamma=d(gamma))


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


@pytest.mark.parametrize('B,ntok', [(2, 4096), (1, 64), (3, 128), (2, 9216)])
def test_st_tail_is_bit_identical_to_the_four_launches(B, ntok):
    """attn2's out-projection in front of the chain (sdmi_k_st_tail): the token stream (updated in place) and the output are the bits of
    the igemm launch + the three launches; statistics as in the ff_tail test."""
    c = _ff_tail_case(B, ntok, 4321 + ntok)
    M, C_ = c['M'], c['C']
    mk = lambda: torch.zeros((B, 32, 8, 16), dtype=torch.int64, device=DEV)
    acc_a, ref_a = mk(), mk()
    out_ref, copy_ref, _, _, _ = _three_launches(c, gn=[(ref_a, 10, 0)])
    t = c['t_prev'].clone()
    out = torch.full((M, C_), float('nan'), device=DEV)
    copy = torch.full((M, C_), float('nan'), dtype=torch.float16, device=DEV)
    K.st_tail(c['ao'], c['wo2'], c['bo2'], t, c['dgamma'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['wpo3'], c['bpo'], c['x_in'], out,
              B, ntok, out_f16=copy, gn=[(acc_a, 10, 0)])
    torch.cuda.synchronize()
    eq = [torch.equal(t, c['t']), torch.equal(out, out_ref), torch.equal(copy, copy_ref)]
    print(f'[st_tail vs launches] t / out / copy equal: {eq}; max diffs {float((t - c["t"]).abs().max()):.3e} '
          f'{float((out - out_ref).abs().max()):.3e}', flush=True)
    assert all(eq), eq
    s, ss = K.gn_acc_sums(acc_a)
    s0, ss0 = K.gn_acc_sums(ref_a)
    assert torch.allclose(s, s0, rtol=1e-6, atol=1e-3) and torch.allclose(ss, ss0, rtol=1e-6, atol=1e-3)


def test_st_tail_repeats_bit_identically_next_to_other_work():
    c = _ff_tail_case(2, 4096, 11)
    out_ref, _, _, _, _ = _three_launches(c, gn=None, want_copy=False)
    junk = torch.empty(96 << 20, device=DEV)          # 384 MB > the 256 MB Infinity Cache
    for i in range(30):
        if i % 2:
            junk.fill_(float(i))
        t = c['t_prev'].clone()
        out = torch.full((c['M'], c['C']), float('nan'), device=DEV)
        K.st_tail(c['ao'], c['wo2'], c['bo2'], t, c['dgamma'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['wpo3'], c['bpo'], c['x_in'], out,
                  2, 4096)
        assert torch.equal(t, c['t']) and torch.equal(out, out_ref), (i, float((out - out_ref).abs().max()))


def _st_head_case(B, ntok, seed, heads=8):
    g = _g(seed)
    C_ = 320
    M = B * ntok
    d = lambda t: t.to(DEV)
    x = torch.randn(B, ntok, C_, generator=g) * 1.7 + 0.4 * torch.randn(1, 1, C_, generator=g)
    gn_g = 1 + 0.2 * torch.randn(C_, generator=g); gn_b = 0.1 * torch.randn(C_, generator=g)
    w_in = torch.randn(C_, C_, generator=g) / math.sqrt(C_); b_in = torch.randn(C_, generator=g) * 0.1
    ln_g = 1 + 0.2 * torch.randn(C_, generator=g); ln_b = 0.1 * torch.randn(C_, generator=g)
    wqkv = (torch.randn(3 * C_, C_, generator=g) / math.sqrt(C_))
    return dict(B=B, ntok=ntok, M=M, C=C_, heads=heads, dh=C_ // heads, x=x, gn_g=gn_g, gn_b=gn_b, w_in=w_in, b_in=b_in, ln_g=ln_g, ln_b=ln_b,
                wqkv=wqkv, dx=d(x), dgn_g=d(gn_g), dgn_b=d(gn_b), w_in3=K.pack_split3(d(w_in)), db_in=d(b_in), dln_g=d(ln_g), dln_b=d(ln_b),
                wqkv16=d(wqkv).half().contiguous())


def _st_head_outputs(c):
    B, ntok, M, C_, heads, dh = c['B'], c['ntok'], c['M'], c['C'], c['heads'], c['dh']
    ntp = (ntok + 7) // 8 * 8
    t = torch.full((M, C_), float('nan'), device=DEV)
    q = torch.full((B * heads, ntok, dh), float('nan'), dtype=torch.float16, device=DEV)
    k = torch.full_like(q, float('nan'))
    vt = torch.zeros((B * heads, dh, ntp), dtype=torch.float16, device=DEV)
    return t, q, k, vt


def _st_head_launches(c):
    """the three launches of the multi-launch path (unet.cpp attn_block): GroupNorm-apply -> proj_in (split-fp16) -> q | k | v"""
    B, ntok, M, C_, heads, dh = c['B'], c['ntok'], c['M'], c['C'], c['heads'], c['dh']
    o = K.groupnorm(c['dx'], None, c['dgn_g'], c['dgn_b'], 1e-6, 0, want=('f16', 'lo'))
    t, q, k, vt = _st_head_outputs(c)
    ln16 = torch.empty(M, C_, dtype=torch.float16, device=DEV)
    part = torch.full((C_ // 32, M, 2), float('nan'), device=DEV)
    K.igemm(o['f16'].view(M, C_), c['w_in3'], C_, B, ntok, 1, ntok, 1, a1=o['lo'].view(M, C_), bias=c['db_in'], out_f32=t, out_f16=ln16,
            f16_scale=c['dln_g'], lnp_out=part, split16=True)
    cs, dn = K.ln_fold_prep(c['wqkv16'], C_, c['dln_g'], c['dln_b'])
    K.igemm(ln16, c['wqkv16'], 3 * C_, B, ntok, 1, ntok, 1, mode=2, lnf=(part, 1e-5, cs, dn),
            heads=dict(segs=[(q, 0), (k, 0), (vt, 1)], heads=heads, dh=dh, ntok=ntok, ntok_pad=vt.shape[2], segC=C_))
    return t, q, k, vt, cs, dn


@pytest.mark.parametrize('B,ntok', [(2, 4096), (1, 64), (3, 128), (2, 9216)])
def test_st_head_is_bit_identical_to_the_three_launches(B, ntok):
    c = _st_head_case(B, ntok, 4321 + ntok)
    M, C_, heads, dh = c['M'], c['C'], c['heads'], c['dh']
    t0, q0, k0, vt0, cs, dn = _st_head_launches(c)
    t, q, k, vt = _st_head_outputs(c)
    K.st_head(c['dx'].view(M, C_), c['dgn_g'], c['dgn_b'], 1e-6, c['w_in3'], c['db_in'], t, c['dln_g'], 1e-5, c['wqkv16'], cs, dn, q, k, vt,
              B, ntok, heads, dh)
    torch.cuda.synchronize()
    # fp32 torch (sanity of the math)
    xn = F.group_norm(c['x'].transpose(1, 2), 32, c['gn_g'], c['gn_b'], 1e-6).transpose(1, 2).reshape(M, C_)
    t_ref = xn @ c['w_in'].t() + c['b_in']
    qkv = F.layer_norm(t_ref, (C_,), c['ln_g'], c['ln_b'], 1e-5) @ c['wqkv'].half().float().t()
    q_ref = qkv[:, :C_].reshape(B, ntok, heads, dh).permute(0, 2, 1, 3).reshape(B * heads, ntok, dh)
    k_ref = qkv[:, C_:2 * C_].reshape(B, ntok, heads, dh).permute(0, 2, 1, 3).reshape(B * heads, ntok, dh)
    v_ref = qkv[:, 2 * C_:].reshape(B, ntok, heads, dh).permute(0, 2, 3, 1).reshape(B * heads, dh, ntok)
    assert K.report(f'st_head t B{B} n{ntok}', t, t_ref, 2e-3) < 2e-3
    assert K.report(f'st_head q B{B} n{ntok}', q.float(), q_ref, 2e-2) < 2e-2
    assert K.report(f'st_head k B{B} n{ntok}', k.float(), k_ref, 2e-2) < 2e-2
    assert K.report(f'st_head vt B{B} n{ntok}', vt[:, :, :ntok].float(), v_ref, 2e-2) < 2e-2
    same = [torch.equal(a, b) for a, b in ((t, t0), (q, q0), (k, k0), (vt, vt0))]
    print(f'[st_head vs launches] t / q / k / vt equal: {same}; max diffs '
          f'{[float((a.float() - b.float()).abs().max()) for a, b in ((t, t0), (q, q0), (k, k0), (vt, vt0))]}', flush=True)
    assert all(same)


def test_st_head_repeats_bit_identically_next_to_other_work():
    c = _st_head_case(2, 4096, 11)
    M, C_, heads, dh = c['M'], c['C'], c['heads'], c['dh']
    cs, dn = K.ln_fold_prep(c['wqkv16'], C_, c['dln_g'], c['dln_b'])
    ref = None
    junk = torch.empty(96 << 20, device=DEV)
    for i in range(30):
        if i % 2:
            junk.fill_(float(i))
        t, q, k, vt = _st_head_outputs(c)
        K.st_head(c['dx'].view(M, C_), c['dgn_g'], c['dgn_b'], 1e-6, c['w_in3'], c['db_in'], t, c['dln_g'], 1e-5, c['wqkv16'], cs, dn, q, k, vt,
                  2, 4096, heads, dh)
        if ref is None:
            ref = (t, q, k, vt)
        else:
            assert all(torch.equal(a, b) for a, b in zip((t, q, k, vt), ref)), i


@pytest.mark.parametrize('B,ntok', [(2, 4096), (1, 64), (3, 128), (2, 9216)])
def test_st_mid_is_bit_identical_to_the_two_launches(B, ntok):
    """out-projection of attn1 (+ residual, in place) and attn2's to_q over norm2 (attention.py:212-213) as one launch"""
    g = _g(977 + ntok)
    C_, heads = 320, 8
    dh, M = C_ // heads, B * ntok
    d = lambda t: t.to(DEV)
    ao = (torch.randn(M, C_, generator=g) * 0.7).half()
    wo = (torch.randn(C_, C_, generator=g) / math.sqrt(C_)).half()
    bo = torch.randn(C_, generator=g) * 0.1
    t_prev = torch.randn(M, C_, generator=g) * 1.5 + 0.3
    ln_g = 1 + 0.2 * torch.randn(C_, generator=g); ln_b = 0.1 * torch.randn(C_, generator=g)
    wq = (torch.randn(C_, C_, generator=g) / math.sqrt(C_)).half()
    cs, dn = K.ln_fold_prep(d(wq), C_, d(ln_g), d(ln_b))
    # the two launches
    t0 = d(t_prev).clone(); ln16 = torch.empty(M, C_, dtype=torch.float16, device=DEV)
    part = torch.full((C_ // 32, M, 2), float('nan'), device=DEV)
    K.igemm(d(ao), d(wo), C_, B, ntok, 1, ntok, 1, bias=d(bo), residual=t0, out_f32=t0, out_f16=ln16, f16_scale=d(ln_g), lnp_out=part)
    q0 = torch.full((B * heads, ntok, dh), float('nan'), dtype=torch.float16, device=DEV)
    K.igemm(ln16, d(wq), C_, B, ntok, 1, ntok, 1, mode=2, lnf=(part, 1e-5, cs, dn),
            heads=dict(segs=[(q0, 0)], heads=heads, dh=dh, ntok=ntok, ntok_pad=(ntok + 7) // 8 * 8, segC=C_))
    # one launch
    t = d(t_prev).clone(); q = torch.full_like(q0, float('nan'))
    K.st_mid(d(ao), d(wo), d(bo), t, d(ln_g), 1e-5, d(wq), cs, dn, q, B, ntok, heads, dh)
    torch.cuda.synchronize()
    t_ref = t_prev + ao.float() @ wo.float().t() + bo
    q_ref = (F.layer_norm(t_ref, (C_,), ln_g, ln_b, 1e-5) @ wq.float().t()).reshape(B, ntok, heads, dh).permute(0, 2, 1, 3).reshape(B * heads, ntok, dh)
    assert K.report(f'st_mid t B{B} n{ntok}', t, t_ref, 2e-3) < 2e-3
    assert K.report(f'st_mid q B{B} n{ntok}', q.float(), q_ref, 2e-2) < 2e-2
    print(f'[st_mid vs launches] t / q equal: {torch.equal(t, t0)} {torch.equal(q, q0)}', flush=True)
    assert torch.equal(t, t0) and torch.equal(q, q0)


@pytest.mark.parametrize('B,ntok,nkv', [(2, 4096, 77), (1, 64, 77), (3, 128, 64), (2, 9216, 77), (2, 256, 128), (1, 32, 5)])
def test_st_mid_with_the_cross_attention_inside_is_bit_identical_to_the_launches(B, ntok, nkv):
    """out-projection of attn1 -> to_q over norm2 -> softmax(q K^T d^-1/2) V over the cached context (attention.py:212-213, 170-193) as ONE launch
    (st_head_kernel CTX) against the st_mid launch + the attention launch: the token stream and the attention output rows bit for bit, and
    the attention output against fp32 torch."""
    g = _g(1977 + ntok + nkv)
    C_, heads = 320, 8
    dh, M = C_ // heads, B * ntok
    nkp = (nkv + 7) // 8 * 8
    d = lambda t: t.to(DEV)
    ao = (torch.randn(M, C_, generator=g) * 0.7).half()
    wo = (torch.randn(C_, C_, generator=g) / math.sqrt(C_)).half()
    bo = torch.randn(C_, generator=g) * 0.1
    t_prev = torch.randn(M, C_, generator=g) * 1.5 + 0.3
    ln_g = 1 + 0.2 * torch.randn(C_, generator=g); ln_b = 0.1 * torch.randn(C_, generator=g)
    wq = (torch.randn(C_, C_, generator=g) / math.sqrt(C_)).half()
    ck = (torch.randn(B * heads, nkv, dh, generator=g) * 1.2).half()
    cv = (torch.randn(B * heads, nkv, dh, generator=g)).half()
    cvt = torch.zeros(B * heads, dh, nkp, dtype=torch.float16); cvt[:, :, :nkv] = cv.transpose(1, 2)
    scale = dh ** -0.5
    cs, dn = K.ln_fold_prep(d(wq), C_, d(ln_g), d(ln_b))
    # the launches: st_mid, then the attention kernel
    t0 = d(t_prev).clone()
    q0 = torch.full((B * heads, ntok, dh), float('nan'), dtype=torch.float16, device=DEV)
    K.st_mid(d(ao), d(wo), d(bo), t0, d(ln_g), 1e-5, d(wq), cs, dn, q0, B, ntok, heads, dh)
    a0 = K.attention(q0, d(ck), d(cvt), heads, nkv, scale).view(M, C_)
    # one launch
    t = d(t_prev).clone()
    a1 = torch.full((M, C_), float('nan'), dtype=torch.float16, device=DEV)
    K.st_mid_ctx(d(ao), d(wo), d(bo), t, d(ln_g), 1e-5, d(wq), cs, dn, d(ck), d(cvt), nkv, scale, a1, B, ntok, heads, dh)
    torch.cuda.synchronize()
    qf = q0.float().cpu()
    ref = torch.softmax(qf @ ck.float().transpose(1, 2) * scale, dim=-1) @ cv.float()
    ref = ref.view(B, heads, ntok, dh).permute(0, 2, 1, 3).reshape(M, C_)
    assert K.report(f'st_mid_ctx attention B{B} n{ntok} nkv{nkv}', a1.float(), ref, 3e-3) < 3e-3
    eq = [torch.equal(t, t0), torch.equal(a1, a0)]
    print(f'[st_mid_ctx vs launches] t / attention output equal: {eq}; max diff {float((a1.float() - a0.float()).abs().max()):.3e}', flush=True)
    assert all(eq), eq


def test_st_mid_ctx_repeats_bit_identically_next_to_other_work():
    g = _g(31)
    B, ntok, nkv, C_, heads = 2, 4096, 77, 320, 8
    dh, M = 40, B * ntok
    d = lambda t: t.to(DEV)
    ao = d((torch.randn(M, C_, generator=g) * 0.7).half()); wo = d((torch.randn(C_, C_, generator=g) / math.sqrt(C_)).half())
    bo = d(torch.randn(C_, generator=g) * 0.1); t_prev = d(torch.randn(M, C_, generator=g) * 1.5 + 0.3)
    ln_g = d(1 + 0.2 * torch.randn(C_, generator=g)); ln_b = d(0.1 * torch.randn(C_, generator=g))
    wq = d((torch.randn(C_, C_, generator=g) / math.sqrt(C_)).half())
    ck = d((torch.randn(B * heads, nkv, dh, generator=g) * 1.2).half())
    cvt = torch.zeros(B * heads, dh, 80, dtype=torch.float16); cvt[:, :, :nkv] = torch.randn(B * heads, dh, nkv, generator=g).half(); cvt = d(cvt)
    cs, dn = K.ln_fold_prep(wq, C_, ln_g, ln_b)
    junk = torch.empty(96 << 20, device=DEV)
    first = None
    for i in range(20):
        if i % 2:
            junk.fill_(float(i))
        t = t_prev.clone(); a1 = torch.full((M, C_), float('nan'), dtype=torch.float16, device=DEV)
        K.st_mid_ctx(ao, wo, bo, t, ln_g, 1e-5, wq, cs, dn, ck, cvt, nkv, dh ** -0.5, a1, B, ntok, heads, dh)
        if first is None:
            first = (t, a1)
        assert torch.equal(t, first[0]) and torch.equal(a1, first[1]), i
