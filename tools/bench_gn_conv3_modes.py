"""Where does gn_conv3 lose inside a UNet call?  Per-launch time of gn_conv3 and of the two launches it replaces (M = 8192, Cin = 320) with
the epilogue variants the UNet uses (GroupNorm statistics targets, fp16 copy) and with cold caches (384 MB rewritten between launches):
    python tools/bench_gn_conv3_modes.py [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import kernels as K  # noqa: E402
import test_gnconv_gpu as T  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, H, W = 2, 64, 64
c = T._case(B, H, W, 320, 0, 5)
N = c['N']
out = torch.empty(B * H * W, N, device='cuda')
copy = torch.empty(B * H * W, N, dtype=torch.float16, device='cuda')
accs = [torch.zeros((B, 32, 8, 16), dtype=torch.int64, device='cuda') for _ in range(2)]
o = K.groupnorm(c['x0'].view(B, H * W, -1), None, c['dgamma'], c['dbeta'], 1e-5, 1)
a16 = o['f16'].view(B * H * W, 320)
junk = torch.empty(96 << 20, device='cuda')


def conv_only(gn, cp):      # the conv launch of the two-launch path alone (operand prepared once)
    K.igemm(a16, c['wp'], N, B, H, W, H, W, ksize=3, bias=c['dbias'], residual=c['dresid'], out_f32=out, out_f16=cp, splitk=1, gn=gn)


def apply_only():
    K.groupnorm(c['x0'].view(B, H * W, -1), None, c['dgamma'], c['dbeta'], 1e-5, 1)


def fused(gn, cp):
    K.gn_conv3(c['x0'], None, c['dgamma'], c['dbeta'], 1e-5, c['wp'], N, out, bias=c['dbias'], residual=c['dresid'], out_f16=cp, gn=gn)


def timeit(fn, cold):
    for _ in range(3):
        fn()
    tot = 0.0
    for i in range(reps):
        if cold:
            junk.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / reps


for name, gn, cp in (('plain', None, None), ('+ 1 statistics target', [(accs[0], 10, 0)], None),
                     ('+ 2 statistics targets', [(accs[0], 10, 0), (accs[1], 30, 640)], None), ('+ 2 targets + fp16 copy', [(accs[0], 10, 0), (accs[1], 30, 640)], copy)):
    for cold in (False, True):
        t_apply = timeit(apply_only, cold)
        t_conv = timeit(lambda: conv_only(gn, cp), cold)
        t_one = timeit(lambda: fused(gn, cp), cold)
        print(f'{name:26s} {"cold" if cold else "hot ":4s}: stats+apply {t_apply:6.1f} us | conv {t_conv:6.1f} us | stats+gn_conv3 {t_one:6.1f} us   (single launches between events)',
              flush=True)
