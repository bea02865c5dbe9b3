"""st_head (one row-strip chain launch) against the three launches it replaces (GroupNorm-apply, split-fp16 proj_in, q|k|v GEMM), on the
bench shape (M = 8192, C = 320):     python tools/bench_st_head.py [reps]
With SDMI_LIB_PATH=<timing library> also prints the phase stamps of wave 0 (cycles, mean over workgroups)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import kernels as K  # noqa: E402
import test_rowchain_gpu as T  # noqa: E402
from stable_diffusion_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = _lib.load()
dbg_fn = getattr(lib, 'sdmi_k_st_head_dbg', None) if 'timing' in os.environ.get('SDMI_LIB_PATH', '') else None
for B, ntok in ((2, 4096), (1, 4096), (2, 9216)):
    c = T._st_head_case(B, ntok, 5)
    M, C_, heads, dh = c['M'], c['C'], c['heads'], c['dh']
    t0, q0, k0, vt0, cs, dn = T._st_head_launches(c)
    t, q, k, vt = T._st_head_outputs(c)

    def one():
        K.st_head(c['dx'].view(M, C_), c['dgn_g'], c['dgn_b'], 1e-6, c['w_in3'], c['db_in'], t, c['dln_g'], 1e-5, c['wqkv16'], cs, dn, q, k, vt,
                  B, ntok, heads, dh)

    def three():
        return T._st_head_launches(c)

    one(); torch.cuda.synchronize()
    print(f'B={B} ntok={ntok} M={M}: bit-identical {[torch.equal(a, b) for a, b in ((t, t0), (q, q0), (k, k0), (vt, vt0))]}', flush=True)
    for rnd in range(3):
        for name, fn in (('three launches (+ stats, prep)', three), ('st_head (+ stats)', one)):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            print(f'  round {rnd} {name:32s} {e0.elapsed_time(e1) * 1e3 / reps:8.1f} us per call', flush=True)
    if dbg_fn is not None and B == 2 and ntok == 4096:
        stamps = torch.zeros((M // 32, 128), dtype=torch.int64, device='cuda')
        dbg_fn.restype = C.c_int; dbg_fn.argtypes = [C.c_void_p]
        dbg_fn(stamps.data_ptr())
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        st = stamps.cpu().double()
        names = ['prologue', 'proj_in', 'proj_in epilogue', 'q', 'q epilogue', 'k', 'k epilogue', 'v', 'v epilogue']
        d = (st[:, 1:10] - st[:, 0:9]).mean(0)
        print('  wave-0 cycles: ' + ' | '.join(f'{n} {v:.0f}' for n, v in zip(names, d)) + f' | total {(st[:, 9] - st[:, 0]).mean():.0f}', flush=True)
        dbg_fn(None)
