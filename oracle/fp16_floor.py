"""CPU emulation of the HIP path's precision design on the oracle: which operand roundings carry the
eps error of a golden case?   python oracle/fp16_floor.py sdv1_real_16x16 [--drop CLASS ...] [--only ...] [--per-layer]
                              python oracle/fp16_floor.py --write-floor          (tests/golden/unet_fp16_floor.json)

The "fp16-operand floor" of a case = the error of the reference arithmetic when every MFMA operand (activations and
weights of the 3x3 convs, the linears and the attention products) is rounded to fp16 ONCE and everything else stays
fp32 -- what north_star's "MFMA fp16" prescribes at best.  tests/test_unet_gpu.py holds the HIP path to that floor on
the cases whose floor itself exceeds the 1e-3 bar (the outlier-statistics family).

Every GEMM operand the HIP path rounds to fp16 is rounded here (activations AND weights, fp32 accumulation = the fp32 CPU
matmul); the classes can be switched to exact (fp32 operands) one at a time to attribute the error.  Not a bit-level model
(summation orders differ): it predicts rms / max-abs to ~10 %.  TEST / ANALYSIS TOOL, imports oracle/ (never the product).
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_ref  # noqa: E402
from oracle.plan import SD_V1, SMALL40, TINY  # noqa: E402
from oracle.weights import make_inputs, make_state_dict  # noqa: E402

CLASSES = ['conv3', 'updown', 'qkv_self', 'q_ctx', 'kv_ctx', 'attn_out', 'geglu', 'ff_out', 'attn_qk', 'attn_pv', 'gn_silu_act', 'ln_act']


def r16(t):
    return t.half().float()


class Emul:
    def __init__(self, on, only_prefix=None, skip_prefixes=()):
        self.on = set(on)
        self.w16 = {}
        self.only_prefix = only_prefix            # round only the GEMMs whose state-dict prefix starts with this
        self.skip_prefixes = tuple(skip_prefixes)  # ... / keep these exact

    def w(self, sd, key):
        if key not in self.w16:
            self.w16[key] = r16(sd[key])
        return self.w16[key]

    def cls_of(self, p):
        if self.only_prefix is not None and not p.startswith(self.only_prefix):
            return None
        if self.skip_prefixes and p.startswith(self.skip_prefixes):
            return None
        if p.endswith(('in_layers.2', 'out_layers.3')):
            return 'conv3'
        if p.endswith(('.op', '.conv')):
            return 'updown'
        if '.attn1.to_q' in p or '.attn1.to_k' in p or '.attn1.to_v' in p:
            return 'qkv_self'
        if '.attn2.to_q' in p:
            return 'q_ctx'
        if '.attn2.to_k' in p or '.attn2.to_v' in p:
            return 'kv_ctx'
        if 'to_out.0' in p:
            return 'attn_out'
        if p.endswith('net.0.proj'):
            return 'geglu'
        if p.endswith('net.2'):
            return 'ff_out'
        return None          # fp32 / split-fp16 on the HIP path: time embedding, emb_layers, conv_in / out, skip / proj_in / proj_out

    def install(self):
        E = self
        orig_conv, orig_lin = unet_ref._conv, unet_ref._lin

        def conv(sd, p, x, stride=1, padding=1):
            c = E.cls_of(p)
            if c in E.on:
                return F.conv2d(r16(x), E.w(sd, p + '.weight'), sd[p + '.bias'], stride=stride, padding=padding)
            return orig_conv(sd, p, x, stride, padding)

        def lin(sd, p, x):
            c = E.cls_of(p)
            if c in E.on:
                return F.linear(r16(x), E.w(sd, p + '.weight'), sd.get(p + '.bias'))
            return orig_lin(sd, p, x)

        def cross_attention(sd, p, x, context, heads):
            q = lin(sd, p + '.to_q', x)
            ctx = x if context is None else context
            k = lin(sd, p + '.to_k', ctx)
            v = lin(sd, p + '.to_v', ctx)
            b, n, c = q.shape
            d = c // heads

            def split(t):
                return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
            q, k, v = split(q), split(k), split(v)
            if 'attn_qk' in E.on:
                q, k = r16(q), r16(k)
            sim = torch.bmm(q, k.transpose(1, 2)) * (d ** -0.5)
            m = sim.amax(dim=-1, keepdim=True)
            pexp = torch.exp(sim - m)
            if 'attn_pv' in E.on:      # P rounded once to fp16 (unnormalised, max 1), v fp16; the denominator from the rounded P (ones row)
                p16, v = r16(pexp), r16(v)
                out = torch.bmm(p16, v) / p16.sum(dim=-1, keepdim=True)
            else:
                out = torch.bmm(pexp, v) / pexp.sum(dim=-1, keepdim=True)
            out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)
            return lin(sd, p + '.to_out.0', out)
        unet_ref._conv, unet_ref._lin, unet_ref.cross_attention = conv, lin, cross_attention
        return orig_conv, orig_lin


def write_floor():
    import glob
    import importlib
    import json
    torch.set_num_threads(os.cpu_count())
    base = [c for c in CLASSES if c not in ('gn_silu_act', 'ln_act')]
    out_path = os.path.join(ROOT, 'tests', 'golden', 'unet_fp16_floor.json')
    doc = json.load(open(out_path)) if os.path.exists(out_path) else {}
    cases = sorted(os.path.basename(f)[5:-4] for f in glob.glob(os.path.join(ROOT, 'tests', 'golden', 'unet_*.npz')))
    key = lambda c: (str(np.load(os.path.join(ROOT, 'tests', 'golden', f'unet_{c}.npz'))['style']) if 'style' in np.load(os.path.join(ROOT, 'tests', 'golden', f'unet_{c}.npz')).files else 'uniform', c.split('_')[0], c)
    sd_key, sd = None, None
    for case in sorted(cases, key=lambda c: (c.split('_')[0],) + tuple(map(str, _wkey(c)))):
        if case in doc or case == 'sdv1_96x96':
            continue
        z = np.load(os.path.join(ROOT, 'tests', 'golden', f'unet_{case}.npz'))
        cfg = {'tiny': TINY, 'small40': SMALL40, 'sdv1': SD_V1}[case.split('_')[0]]
        style = str(z['style']) if 'style' in z.files else 'uniform'
        k = (case.split('_')[0], int(z['weight_seed']), style)
        if k != sd_key:
            sd_key, sd = k, make_state_dict(cfg, int(z['weight_seed']), style=style)
        x, t, ctx = make_inputs(cfg, int(z['batch']), int(z['h']), int(z['w']), seed=int(z['input_seed']), ctx_len=int(z['ctx_len']),
                                timesteps=tuple(int(v) for v in z['t']), style=style)
        importlib.reload(unet_ref)
        Emul(base).install()
        err = (unet_ref.unet_forward(sd, cfg, x, t, ctx) - torch.from_numpy(z['eps'])).abs()
        doc[case] = {'maxabs': float(err.max()), 'rms': float(err.pow(2).mean().sqrt())}
        print(case, doc[case], flush=True)
        with open(out_path, 'w') as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    importlib.reload(unet_ref)


def _wkey(case):
    z = np.load(os.path.join(ROOT, 'tests', 'golden', f'unet_{case}.npz'))
    return (int(z['weight_seed']), str(z['style']) if 'style' in z.files else 'uniform')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('case', nargs='?')
    ap.add_argument('--write-floor', action='store_true', help='floor statistics of every SD-v1 golden -> tests/golden/unet_fp16_floor.json')
    ap.add_argument('--drop', nargs='*', default=None, help='classes made exact one at a time (default: every class)')
    ap.add_argument('--only', nargs='*', default=None, help='classes rounded one at a time, everything else exact')
    ap.add_argument('--drop-together', nargs='*', action='append', default=[], help='a set of classes made exact together (repeatable)')
    ap.add_argument('--per-layer', action='store_true', help='round ONE layer (ResBlock / SpatialTransformer / resampler) at a time')
    ap.add_argument('--skip', nargs='*', default=None, help='layer prefixes kept exact while everything else is rounded')
    a = ap.parse_args()
    if a.write_floor:
        return write_floor()
    z = np.load(os.path.join(ROOT, 'tests', 'golden', f'unet_{a.case}.npz'))
    cfg = {'tiny': TINY, 'small40': SMALL40, 'sdv1': SD_V1}[a.case.split('_')[0]]
    style = str(z['style']) if 'style' in z.files else 'uniform'
    sd = make_state_dict(cfg, int(z['weight_seed']), style=style)
    x, t, ctx = make_inputs(cfg, int(z['batch']), int(z['h']), int(z['w']), seed=int(z['input_seed']), ctx_len=int(z['ctx_len']),
                            timesteps=tuple(int(v) for v in z['t']), style=style)
    ref = torch.from_numpy(z['eps'])
    torch.set_num_threads(os.cpu_count())
    base = [c for c in CLASSES if c not in ('gn_silu_act', 'ln_act')]

    def run(on, label, only_prefix=None, skip_prefixes=()):
        import importlib
        importlib.reload(unet_ref)
        e = Emul(on, only_prefix, skip_prefixes)
        e.install()
        eps = unet_ref.unet_forward(sd, cfg, x, t, ctx)
        err = (eps - ref).abs()
        print(f'{label:28s} max-abs {float(err.max()):.3e}  rms {float(err.pow(2).mean().sqrt()):.3e}', flush=True)
        return float(err.pow(2).mean())
    if a.per_layer:
        from oracle.plan import build_plan
        tot = run(base, 'all classes rounded')
        rows = []
        for L in build_plan(cfg).all_layers():
            if L.kind in ('res', 'attn', 'down', 'up'):
                v = run(base, 'only ' + L.prefix, only_prefix=L.prefix + '.')
                rows.append((v, L.prefix, L.kind))
        s_ = sum(v for v, _, _ in rows)
        print(f'sum of per-layer variances / total variance = {s_ / tot:.2f}')
        for v, pfx, kind in sorted(rows, reverse=True)[:16]:
            print(f'  {pfx:24s} {kind:5s} {v / s_ * 100:5.1f} % of the summed variance')
        if a.skip:
            run(base, 'all but ' + ' '.join(a.skip), skip_prefixes=tuple(p_ + '.' for p_ in a.skip))
        return
    if a.skip:
        run(base, 'all classes rounded')
        run(base, 'all but ' + ' '.join(a.skip), skip_prefixes=tuple(p_ + '.' for p_ in a.skip))
        return
    run([], 'exact (oracle)')
    run(base, 'all classes rounded')
    if a.only is not None:
        for c in (a.only or base):
            run([c], f'only {c}')
    for group in a.drop_together:
        run([k for k in base if k not in group], 'all but ' + '+'.join(group))
    if a.drop_together and not a.drop:
        return
    for c in (a.drop if a.drop else ([] if a.only is not None else base)):
        run([k for k in base if k != c], f'all but {c}')


if __name__ == '__main__':
    main()
