"""ff_tail (one row-strip chain launch) against the three launches it replaces, on the bench shape (M = 8192, C = 320):
    python tools/bench_ff_tail.py [reps]
Interleaved rounds, CUDA events around `reps` back-to-back calls of each path (the weights stay hot: the in-UNet figure is the
per-shape profile's).  Also prints whether the outputs are bit-identical."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import kernels as K  # noqa: E402
import test_rowchain_gpu as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for B, ntok in ((2, 4096), (1, 4096), (4, 4096), (2, 9216)):
    c = T._ff_tail_case(B, ntok, 5)
    M, C_ = c['M'], c['C']
    out = torch.empty(M, C_, device='cuda'); copy = torch.empty(M, C_, dtype=torch.float16, device='cuda')

    def one():
        K.ff_tail(c['ln16'], c['part'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['t'], c['wpo3'], c['bpo'], c['x_in'], out, B, ntok,
                  out_f16=copy)

    def three():
        return T._three_launches(c)

    ref = three()[0]; one(); torch.cuda.synchronize()
    print(f'B={B} ntok={ntok} M={M}: bit-identical {torch.equal(out, ref)}', flush=True)
    for rnd in range(3):
        for name, fn in (('three launches', three), ('ff_tail', one)):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            flops = 2.0 * M * (8 + 4 + 1) * C_ * C_
            print(f'  round {rnd} {name:15s} {us:8.1f} us per call   {flops / us * 1e-6:7.1f} TFLOP/s (algorithmic)', flush=True)
