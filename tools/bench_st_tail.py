"""st_tail (attn2's out-projection + the ff_tail chain as ONE launch) against out-projection launch + ff_tail launch, bench shape
(M = 8192, C = 320):   python tools/bench_st_tail.py [reps]      Interleaved rounds, CUDA events around `reps` back-to-back calls."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import kernels as K  # noqa: E402
import test_rowchain_gpu as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for B, ntok in ((2, 4096),):
    c = T._ff_tail_case(B, ntok, 5)
    M, C_ = c['M'], c['C']
    out = torch.empty(M, C_, device='cuda'); copy = torch.empty(M, C_, dtype=torch.float16, device='cuda')
    t = c['t_prev'].clone(); ln16 = torch.empty_like(c['ln16']); part = torch.empty_like(c['part'])

    def ff():
        K.ff_tail(c['ln16'], c['part'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['t'], c['wpo3'], c['bpo'], c['x_in'], out, B, ntok,
                  out_f16=copy)

    def two():
        K.igemm(c['ao'], c['wo2'], C_, B, ntok, 1, ntok, 1, bias=c['bo2'], residual=c['t_prev'], out_f32=t, out_f16=ln16, f16_scale=c['dgamma'],
                lnp_out=part)
        K.ff_tail(ln16, part, 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], t, c['wpo3'], c['bpo'], c['x_in'], out, B, ntok, out_f16=copy)

    def one():      # (t accumulates across calls: timing only)
        K.st_tail(c['ao'], c['wo2'], c['bo2'], t, c['dgamma'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['wpo3'], c['bpo'], c['x_in'], out,
                  B, ntok, out_f16=copy)

    for rnd in range(3):
        for name, fn in (('ff_tail alone', ff), ('out-proj + ff_tail', two), ('st_tail', one)):
            t.copy_(c['t_prev'])
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            print(f'  round {rnd} {name:20s} {e0.elapsed_time(e1) * 1e3 / reps:8.1f} us per call', flush=True)
