"""gn_conv3 (GroupNorm + SiLU + conv3x3 as ONE launch) against GroupNorm-apply + conv launches, bench shapes of the 64 x 64 level:
    python tools/bench_gn_conv3.py [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import kernels as K  # noqa: E402
import test_gnconv_gpu as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for B, H, W, c0, c1 in ((2, 64, 64, 320, 0), (2, 64, 64, 320, 320), (2, 64, 64, 640, 320)):
    c = T._case(B, H, W, c0, c1, 5)
    N = c['N']
    out = torch.empty(B * H * W, N, device='cuda')

    def two():
        return T._two_launches(c, True)

    def one():
        K.gn_conv3(c['x0'], c['x1'], c['dgamma'], c['dbeta'], 1e-5, c['wp'], N, out, bias=c['dbias'], residual=c['dresid'])

    ref = two()[0]; one(); torch.cuda.synchronize()
    print(f'B={B} {H}x{W} Cin={c0}+{c1}: bit-identical {torch.equal(out, ref)}', flush=True)
    for rnd in range(3):
        for name, fn in (('two launches (+ stats)', two), ('gn_conv3 (+ stats)', one)):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            print(f'  round {rnd} {name:24s} {e0.elapsed_time(e1) * 1e3 / reps:8.1f} us per call', flush=True)
