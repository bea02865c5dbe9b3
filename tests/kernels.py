"""torch-tensor wrappers around the kernel-level C-ABI entry points (sdmi_k_*), shared by the GPU tests."""
import ctypes as C

import torch

from stable_diffusion_amd import _lib


def _s():
    return _lib.stream_ptr()


def pack_conv_weight(w):
    """[O,I,KH,KW] fp32 cuda -> fp16 [O, KH*KW*I]"""
    O, I, KH, KW = w.shape
    dst = torch.empty((O, KH * KW * I), dtype=torch.float16, device=w.device)
    _lib.check(_lib.load().sdmi_k_pack_conv_weight(w.contiguous().data_ptr(), dst.data_ptr(), O, I, KH, KW, _s()))
    return dst


def pack_geglu(w, b):
    N, K = w.shape
    wd = torch.empty((N, K), dtype=torch.float16, device=w.device)
    bd = torch.empty((N,), dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().sdmi_k_pack_geglu(w.contiguous().data_ptr(), b.contiguous().data_ptr(), wd.data_ptr(),
                                             bd.data_ptr(), N, K, _s()))
    return wd, bd


def pack_split3(w):
    N, K_ = w.shape
    dst = torch.empty((N, 3 * K_), dtype=torch.float16, device=w.device)
    _lib.check(_lib.load().sdmi_k_pack_split3(w.contiguous().data_ptr(), dst.data_ptr(), N, K_, _s()))
    return dst


def cast_f16(x, want_lo=False):
    hi = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    lo = torch.empty_like(hi) if want_lo else None
    _lib.check(_lib.load().sdmi_k_cast_f16(x.data_ptr(), hi.data_ptr(), _lib.ptr(lo), x.numel(), _s()))
    return hi, lo


def igemm(a0, w, N, B, Hin, Win, Hout, Wout, ksize=1, stride=1, up=0, a1=None, a2=None, bias=None, rowvec=None,
          residual=None, out_f32=None, out_f16=None, ldo=None, mode=0, splitk=1, tile=-1, dma=-1, heads=None, fused_splitk=True,
          asym_pad=0, gn=None, split16=False, f16_scale=None, lnp_out=None, lnf=None, pgn=None, out_lo=None):
    """a0/a1: fp16 [B*Hin*Win, C] ; w: fp16 [N, K].
    LayerNorm fold: producer f16_scale (gamma [N]) + lnp_out ([N/32, M, 2] fp32); consumer lnf = (partials, eps, cs, d)."""
    d = _lib.IGemmDesc()
    d.a0 = a0.data_ptr(); d.c0 = a0.shape[1]; d.lda0 = a0.stride(0)
    if a1 is not None:
        d.a1 = a1.data_ptr(); d.c1 = a1.shape[1]; d.lda1 = a1.stride(0)
    if a2 is not None:
        d.a2 = a2.data_ptr(); d.c2 = a2.shape[1]; d.lda2 = a2.stride(0)
    d.B, d.Hin, d.Win, d.Hout, d.Wout, d.ksize, d.stride, d.up = B, Hin, Win, Hout, Wout, ksize, stride, up
    d.w = w.data_ptr(); d.N = N; d.mode = mode
    d.bias = _lib.ptr(bias)
    if rowvec is not None:
        d.rowvec = rowvec.data_ptr(); d.ld_rowvec = rowvec.stride(0)
    if residual is not None:
        d.residual = residual.data_ptr(); d.ldr = residual.stride(0)
    d.out_f32 = _lib.ptr(out_f32); d.out_f16 = _lib.ptr(out_f16)
    d.ldo = ldo if ldo is not None else (out_f32.stride(0) if out_f32 is not None else
                                         (out_f16.stride(0) if out_f16 is not None else 0))
    if heads is not None:
        for i, (t, kind) in enumerate(heads['segs']):
            d.seg_dst[i] = t.data_ptr(); d.seg_kind[i] = kind
        d.heads, d.dh, d.ntok, d.ntok_pad, d.segC = heads['heads'], heads['dh'], heads['ntok'], heads['ntok_pad'], heads['segC']
    d.splitk, d.tile, d.dma = splitk, tile, dma
    d.asym_pad = asym_pad
    d.split16 = 1 if split16 else 0      # a0 = hi, a1 = lo, w = pack_split3 ([N][3 c0])
    d.f16_scale = _lib.ptr(f16_scale); d.lnp_out = _lib.ptr(lnp_out)
    d.out_lo = _lib.ptr(out_lo)
    if lnf is not None:
        part, eps, cs, dn = lnf
        d.lnf_part = part.data_ptr(); d.lnf_npart = part.shape[0]; d.lnf_eps = float(eps)
        d.lnf_cs = cs.data_ptr(); d.lnf_d = dn.data_ptr()
    if gn:      # [(acc int64 tensor [B,32,8,16] (zeroed), cpg, cbase)]
        d.gn_n = len(gn)
        for i, (acc, cpg, cbase) in enumerate(gn):
            d.gn_acc[i] = acc.data_ptr(); d.gn_cpg[i] = cpg; d.gn_cbase[i] = cbase
    applied = C.c_int32(0)
    if pgn is not None:    # (gamma, beta, eps, silu, out fp16 [M, N], keep_f32): GroupNorm (+ SiLU) inside the split-K reduction
        ga, be, eps, silu, o16, keep = pgn
        d.pgn_gamma = ga.data_ptr(); d.pgn_beta = be.data_ptr(); d.pgn_eps = float(eps); d.pgn_silu = int(silu)
        d.pgn_out = o16.data_ptr(); d.pgn_keep_f32 = int(keep); d.pgn_applied = C.pointer(applied)
    ws = None
    if splitk != 1:
        M = B * Hout * Wout
        ws = torch.empty((16 * (M + 255) * (N + 255),), dtype=torch.float32, device=a0.device)
        d.splitk_ws = ws.data_ptr(); d.splitk_ws_floats = ws.numel()
        if fused_splitk:
            cnt = _splitk_counters(a0.device)
            d.splitk_cnt = cnt.data_ptr(); d.splitk_cnt_ints = cnt.numel()
    _lib.check(_lib.load().sdmi_k_igemm(C.byref(d), _s()))
    return int(applied.value)


def ff_tail(ln16, part, eps, csd, wgg, wff2, bff2, t, wpo3, bpo, x_in, out_f32, B, ntok, out_f16=None, gn=None):
    """GEGLU -> FF-out -> proj_out as one launch (sdmi_k_ff_tail): out = x_in + proj_out(t + FF(norm3(t))).
    ln16 [M, C] fp16(gamma3 * t), part [C/32, M, 2]; csd [4, 2 * 2C] per-chunk {cs | d}; wpo3 = pack_split3(w_proj_out)."""
    C_ = ln16.shape[1]
    d = _lib.IGemmDesc()
    d.c0 = C_; d.lda0 = C_; d.B, d.Hin, d.Win, d.Hout, d.Wout, d.ksize, d.stride = B, ntok, 1, ntok, 1, 1, 1
    d.w = wpo3.data_ptr(); d.N = C_; d.mode = 0; d.split16 = 1; d.splitk = 1; d.tile = -1; d.dma = -1
    d.bias = _lib.ptr(bpo); d.residual = x_in.data_ptr(); d.ldr = x_in.stride(0)
    d.out_f32 = out_f32.data_ptr(); d.out_f16 = _lib.ptr(out_f16); d.ldo = out_f32.stride(0)
    if gn:
        d.gn_n = len(gn)
        for i, (acc, cpg, cbase) in enumerate(gn):
            d.gn_acc[i] = acc.data_ptr(); d.gn_cpg[i] = cpg; d.gn_cbase[i] = cbase
    _lib.check(_lib.load().sdmi_k_ff_tail(C.byref(d), ln16.data_ptr(), part.data_ptr(), float(eps), csd.data_ptr(), wgg.data_ptr(),
                                          wff2.data_ptr(), bff2.data_ptr(), t.data_ptr(), _s()))


def st_tail(a16, wo16, bo, t, ln_gamma, eps, csd, wgg, wff2, bff2, wpo3, bpo, x_in, out_f32, B, ntok, out_f16=None, gn=None):
    """attn2's out-projection -> GEGLU -> FF-out -> proj_out as one launch (sdmi_k_st_tail): t += a16 Wo^T + bo in place, then
    out = x_in + proj_out(t + FF(norm3(t)))."""
    C_ = a16.shape[1]
    d = _lib.IGemmDesc()
    d.c0 = C_; d.lda0 = C_; d.B, d.Hin, d.Win, d.Hout, d.Wout, d.ksize, d.stride = B, ntok, 1, ntok, 1, 1, 1
    d.w = wpo3.data_ptr(); d.N = C_; d.mode = 0; d.split16 = 1; d.splitk = 1; d.tile = -1; d.dma = -1
    d.bias = _lib.ptr(bpo); d.residual = x_in.data_ptr(); d.ldr = x_in.stride(0)
    d.out_f32 = out_f32.data_ptr(); d.out_f16 = _lib.ptr(out_f16); d.ldo = out_f32.stride(0)
    if gn:
        d.gn_n = len(gn)
        for i, (acc, cpg, cbase) in enumerate(gn):
            d.gn_acc[i] = acc.data_ptr(); d.gn_cpg[i] = cpg; d.gn_cbase[i] = cbase
    _lib.check(_lib.load().sdmi_k_st_tail(C.byref(d), a16.data_ptr(), wo16.data_ptr(), bo.data_ptr(), t.data_ptr(), ln_gamma.data_ptr(),
                                          float(eps), csd.data_ptr(), wgg.data_ptr(), wff2.data_ptr(), bff2.data_ptr(), _s()))


def gn_conv3(x0, x1, gamma, beta, eps, wp, N, out_f32, bias=None, rowvec=None, residual=None, out_f16=None, gn=None):
    """conv3x3(SiLU(GroupNorm32(cat(x0, x1)))) as ONE launch (sdmi_k_gn_conv3).  x0 / x1: fp32 [B, H, W, C]; wp = pack_conv_weight(w);
    out_f32 [B * H * W, N]."""
    B, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    d = _lib.IGemmDesc()
    d.c0 = c0 + c1; d.lda0 = c0 + c1
    d.B, d.Hin, d.Win, d.Hout, d.Wout, d.ksize, d.stride, d.up = B, H, W, H, W, 3, 1, 0
    d.w = wp.data_ptr(); d.N = N; d.mode = 0; d.splitk = 1; d.tile = -1; d.dma = -1
    d.bias = _lib.ptr(bias)
    if rowvec is not None:
        d.rowvec = rowvec.data_ptr(); d.ld_rowvec = rowvec.stride(0)
    if residual is not None:
        d.residual = residual.data_ptr(); d.ldr = residual.stride(0)
    d.out_f32 = out_f32.data_ptr(); d.out_f16 = _lib.ptr(out_f16); d.ldo = out_f32.stride(0)
    if gn:
        d.gn_n = len(gn)
        for i, (acc, cpg, cbase) in enumerate(gn):
            d.gn_acc[i] = acc.data_ptr(); d.gn_cpg[i] = cpg; d.gn_cbase[i] = cbase
    n = _lib.load().sdmi_k_groupnorm_ws_floats(B, H * W)
    ws = torch.empty((n,), dtype=torch.float32, device=x0.device)
    _lib.check(_lib.load().sdmi_k_gn_conv3(C.byref(d), x0.data_ptr(), _lib.ptr(x1), c0, c1, ws.data_ptr(), n, gamma.data_ptr(),
                                           beta.data_ptr(), float(eps), _s()))


def st_head(x, gn_gamma, gn_beta, gn_eps, w_in3, b_in, t, ln_gamma, ln_eps, wqkv, cs, dn, q, k, vt, B, ntok, heads, dh):
    """GroupNorm-apply -> proj_in -> q | k | v as one launch (sdmi_k_st_head).  x [B * ntok, C] fp32; w_in3 = pack_split3(proj_in weight);
    wqkv [3C, C] fp16; (cs, dn) = ln_fold_prep(wqkv, C, norm1 weight, norm1 bias); t [M, C] fp32, q / k [B * heads, ntok, dh],
    vt [B * heads, dh, ntok_pad] fp16 are written."""
    C_ = x.shape[1]
    n = _lib.load().sdmi_k_groupnorm_ws_floats(B, ntok)
    ws = torch.empty((n,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().sdmi_k_st_head(x.data_ptr(), ws.data_ptr(), n, gn_gamma.data_ptr(), gn_beta.data_ptr(), float(gn_eps),
                                          w_in3.data_ptr(), b_in.data_ptr(), t.data_ptr(), ln_gamma.data_ptr(), float(ln_eps),
                                          wqkv.data_ptr(), cs.data_ptr(), dn.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(),
                                          B, ntok, vt.shape[2], heads, dh, C_, _s()))


def st_mid(a16, wo16, bo, t, ln_gamma, ln_eps, wq16, cs, dn, q, B, ntok, heads, dh):
    """t += a16 Wo^T + bo (in place), q = norm2(t) Wq^T per head, as one launch (sdmi_k_st_mid)"""
    _lib.check(_lib.load().sdmi_k_st_mid(a16.data_ptr(), wo16.data_ptr(), bo.data_ptr(), t.data_ptr(), ln_gamma.data_ptr(), float(ln_eps),
                                         wq16.data_ptr(), cs.data_ptr(), dn.data_ptr(), q.data_ptr(), B, ntok, heads, dh, a16.shape[1], _s()))


def st_mid_ctx(a16, wo16, bo, t, ln_gamma, ln_eps, wq16, cs, dn, ck, cvt, nkv, scale, ao_out, B, ntok, heads, dh):
    """st_mid with the cross-attention behind to_q inside the launch (sdmi_k_st_mid_ctx): ck [B * heads, nkv, dh], cvt [B * heads, dh, nkv_pad]"""
    _lib.check(_lib.load().sdmi_k_st_mid_ctx(a16.data_ptr(), wo16.data_ptr(), bo.data_ptr(), t.data_ptr(), ln_gamma.data_ptr(), float(ln_eps),
                                             wq16.data_ptr(), cs.data_ptr(), dn.data_ptr(), ck.data_ptr(), cvt.data_ptr(), nkv, cvt.shape[2],
                                             float(scale), ao_out.data_ptr(), B, ntok, heads, dh, a16.shape[1], _s()))


_CNT = {}


def _splitk_counters(dev):
    """tile counters of the fused split-K reduction: zero once, the kernels leave them zero"""
    if dev not in _CNT:
        _CNT[dev] = torch.zeros((8192,), dtype=torch.int32, device=dev)
    return _CNT[dev]


def gn_acc_sums(acc):
    """accumulator words [B, 32, 8 slots, 16 (4 used: one 128-byte line per slot)] int64 -> (sum, sumsq) [B, 32] float64"""
    a = acc.cpu().to(torch.float64)
    s = (a[..., 0] + a[..., 1] / 2.0 ** 40).sum(-1)
    ss = (a[..., 2] + a[..., 3] / 2.0 ** 40).sum(-1)
    return s, ss


def attention_ctx(x16, wq16, k, vt, heads, nkv, scale, lnf=None):
    """x16 [B*nq, C] fp16, wq16 [C, C] fp16, k [BH, nkv, d], vt [BH, d, nkv_pad] -> [B, nq, C] fp16 (to_q inside the kernel);
    lnf = (partials [C/32, B*nq, 2], eps, cs, d): LayerNorm fold"""
    BH, _, d = k.shape
    B = BH // heads
    nq = x16.shape[0] // B
    out = torch.empty((B, nq, heads * d), dtype=torch.float16, device=x16.device)
    part, eps, cs, dn = lnf if lnf is not None else (None, 1e-5, None, None)
    _lib.check(_lib.load().sdmi_k_attention_ctx(x16.data_ptr(), wq16.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), BH, heads,
                                                nq, nkv, vt.shape[2], d, float(scale), _lib.ptr(part), float(eps), _lib.ptr(cs),
                                                _lib.ptr(dn), _s()))
    return out


def ln_fold_prep(w16, K, gamma, beta, bias=None):
    """column terms of a LayerNorm-folding GEMM from the packed fp16 weights w16 [N, ldw >= K]: (cs, d) fp32 [N]"""
    N = w16.shape[0]
    cs = torch.empty((N,), dtype=torch.float32, device=w16.device)
    dn = torch.empty((N,), dtype=torch.float32, device=w16.device)
    _lib.check(_lib.load().sdmi_k_ln_fold_prep(w16.data_ptr(), N, K, w16.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                               _lib.ptr(bias), cs.data_ptr(), dn.data_ptr(), _s()))
    return cs, dn


def attention(q, k, vt, heads, nkv, scale):
    """q [BH,nq,d], k [BH,nkv,d], vt [BH,d,nkv_pad] fp16 -> [B, nq, heads*d] fp16"""
    BH, nq, d = q.shape
    out = torch.empty((BH // heads, nq, heads * d), dtype=torch.float16, device=q.device)
    _lib.check(_lib.load().sdmi_k_attention(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), BH, heads, nq,
                                            nkv, vt.shape[2], d, float(scale), _s()))
    return out


def attention_causal(q, k, vt, heads, scale):
    """q, k [BH,n,d], vt [BH,d,n_pad] fp16 -> [B, n, heads*d] fp16, query i sees keys <= i"""
    BH, n, d = q.shape
    out = torch.empty((BH // heads, n, heads * d), dtype=torch.float16, device=q.device)
    _lib.check(_lib.load().sdmi_k_attention_causal(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), BH, heads, n,
                                                   vt.shape[2], d, float(scale), _s()))
    return out


def groupnorm(x0, x1, gamma, beta, eps, silu, want=('f16',)):
    """x0/x1: fp32 [B, HW, C] -> dict of outputs"""
    B, HW, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[2]
    C_ = c0 + c1
    dev = x0.device
    o16 = torch.empty((B, HW, C_), dtype=torch.float16, device=dev) if 'f16' in want else None
    o32 = torch.empty((B, HW, C_), dtype=torch.float32, device=dev) if 'f32' in want else None
    raw = torch.empty((B, HW, C_), dtype=torch.float16, device=dev) if 'raw' in want else None
    olo = torch.empty((B, HW, C_), dtype=torch.float16, device=dev) if 'lo' in want else None
    rlo = torch.empty((B, HW, C_), dtype=torch.float16, device=dev) if 'raw_lo' in want else None
    n = _lib.load().sdmi_k_groupnorm_ws_floats(B, HW)
    ws = torch.empty((n,), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().sdmi_k_groupnorm(x0.data_ptr(), _lib.ptr(x1), c0, c1, B, HW, gamma.data_ptr(),
                                            beta.data_ptr(), float(eps), int(silu), _lib.ptr(o16), _lib.ptr(o32),
                                            _lib.ptr(raw), _lib.ptr(olo), _lib.ptr(rlo), ws.data_ptr(), n, _s()))
    return dict(f16=o16, f32=o32, raw=raw, lo=olo, raw_lo=rlo)


def layernorm(x, gamma, beta, eps=1e-5):
    M, C_ = x.shape
    out = torch.empty((M, C_), dtype=torch.float16, device=x.device)
    _lib.check(_lib.load().sdmi_k_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), M, C_,
                                            float(eps), _s()))
    return out


def timestep_embedding(t, dim):
    B = t.shape[0]
    out = torch.empty((B, dim), dtype=torch.float32, device=t.device)
    if t.dtype == torch.int64:
        _lib.check(_lib.load().sdmi_k_timestep_embedding(t.data_ptr(), None, out.data_ptr(), B, dim, _s()))
    else:
        _lib.check(_lib.load().sdmi_k_timestep_embedding(None, t.float().contiguous().data_ptr(), out.data_ptr(), B, dim, _s()))
    return out


def small_linear(x, w, b, silu_in):
    B, K = x.shape
    N = w.shape[0]
    out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().sdmi_k_small_linear(x.data_ptr(), x.stride(0), w.data_ptr(), _lib.ptr(b), out.data_ptr(), N,
                                               B, N, K, int(silu_in), _s()))
    return out


def conv_in(x, w, b):
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    out = torch.empty((B, H * W, Cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().sdmi_k_conv_in(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, Cin, H, W, Cout, _s()))
    return out


def conv_out(h, w, b, B, H, W):
    """h: fp32 [B, HW, Cin]; w: OIHW fp32"""
    Cout, Cin = w.shape[0], w.shape[1]
    wp = torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device=h.device)
    _lib.check(_lib.load().sdmi_k_pack_conv_out(w.contiguous().data_ptr(), wp.data_ptr(), Cout, Cin, _s()))
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=h.device)
    _lib.check(_lib.load().sdmi_k_conv_out(h.data_ptr(), wp.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, _s()))
    return out


def pointwise_nchw(x, w, b, in_scale=1.0):
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().sdmi_k_pointwise_nchw(x.data_ptr(), w.contiguous().data_ptr(), _lib.ptr(b), out.data_ptr(), B,
                                                 Cin, Cout, H * W, float(in_scale), _s()))
    return out


def softmax_rows(S, scale):
    rows, cols = S.shape
    P = torch.empty((rows, cols), dtype=torch.float16, device=S.device)
    _lib.check(_lib.load().sdmi_k_softmax_rows(S.data_ptr(), P.data_ptr(), rows, cols, float(scale), _s()))
    return P


def sampler_step(eps_model, cfg, scale, x, mode, old, a_t, a_prev, sigma, s1m, noise=None):
    e_t = torch.empty_like(x); x_prev = torch.empty_like(x); pred = torch.empty_like(x)
    o = list(old) + [None] * 3
    _lib.check(_lib.load().sdmi_sampler_step(eps_model.data_ptr(), int(cfg), float(scale), x.data_ptr(), mode,
                                             _lib.ptr(o[0]), _lib.ptr(o[1]), _lib.ptr(o[2]), float(a_t), float(a_prev),
                                             float(sigma), float(s1m), _lib.ptr(noise), e_t.data_ptr(),
                                             x_prev.data_ptr(), pred.data_ptr(), x.numel(), _s()))
    return e_t, x_prev, pred


def report(name, got, ref, tol):
    """max-abs / rel diagnostics printed for the GPU log; returns the max-abs error."""
    got = got.float().cpu(); ref = ref.float().cpu()
    d = (got - ref).abs()
    mx = d.max().item()
    idx = int(d.flatten().argmax())
    print(f'[{name}] max-abs {mx:.3e} (tol {tol:.1e}) ref-absmax {ref.abs().max().item():.3e} rms-err '
          f'{d.pow(2).mean().sqrt().item():.3e} at flat {idx}: got {got.flatten()[idx].item():.6f} ref '
          f'{ref.flatten()[idx].item():.6f} nan={bool(torch.isnan(got).any())}', flush=True)
    return mx


def conv3gn(x0, x1, gamma, beta, eps, w, bias=None, rowvec=None, residual=None, splitk=0, tile=-1, want_raw=False):
    """GroupNorm + SiLU folded into the halo-staged 3x3 conv.  x0/x1: fp32 [B, H, W, C]; w: OIHW fp32 -> out fp32 [B*H*W, N]
    (and, with want_raw, the split-fp16 copy (hi, lo) of cat(x0, x1))"""
    B, H, W, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    N = w.shape[0]
    dev = x0.device
    wp = pack_conv_weight(w)
    out = torch.full((B * H * W, N), float('nan'), device=dev)
    n = _lib.load().sdmi_k_groupnorm_ws_floats(B, H * W)
    gws = torch.empty((n,), dtype=torch.float32, device=dev)
    ws = torch.empty((16 * B * H * W * N,), dtype=torch.float32, device=dev)
    rhi = torch.full((B * H * W, c0 + c1), float('nan'), dtype=torch.float16, device=dev) if want_raw else None
    rlo = torch.full((B * H * W, c0 + c1), float('nan'), dtype=torch.float16, device=dev) if want_raw else None
    _lib.check(_lib.load().sdmi_k_conv3gn(
        x0.data_ptr(), _lib.ptr(x1), c0, c1, B, H, W, gamma.data_ptr(), beta.data_ptr(), float(eps), wp.data_ptr(), N,
        _lib.ptr(bias), _lib.ptr(rowvec), 0 if rowvec is None else rowvec.stride(0), _lib.ptr(residual),
        0 if residual is None else residual.stride(0), out.data_ptr(), N, splitk, ws.data_ptr(), ws.numel(),
        gws.data_ptr(), n, tile, _lib.ptr(rhi), _lib.ptr(rlo), _s()))
    return (out, rhi, rlo) if want_raw else out
