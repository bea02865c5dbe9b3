"""Where does ff_tail_kernel spend its cycles?  Needs the timing build:
    bash tools/build_rc_timing.sh && SDMI_LIB_PATH=stable-diffusion_amd/libsdmi_rctiming.so python tools/ff_tail_timing.py
Per ablation (0 = the product kernel, 1 = stream + barriers only, 2 = compute only, 3 = no GEGLU / FF-out epilogue arithmetic, 4 = no
barriers): launch time (CUDA events) and, from wave 0's s_memtime stamps of every workgroup, the cycles of each phase: prologue, the
k-tile steps (two units each) of the GEGLU passes / FF-out partials per hidden chunk, the FF-out epilogue, proj_out, the final epilogue."""
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

lib = _lib.load()
dbg_fn = getattr(lib, 'sdmi_k_ff_tail_dbg')
dbg_fn.restype = C.c_int
dbg_fn.argtypes = [C.c_void_p, C.c_int]
B, ntok = 2, 4096
c = T._ff_tail_case(B, ntok, 5)
M, C_ = c['M'], c['C']
nwg = M // 32
out = torch.empty(M, C_, device='cuda')
stamps = torch.zeros(nwg, 128, dtype=torch.int64, device='cuda')


def one():
    K.ff_tail(c['ln16'], c['part'], 1e-5, c['csd'], c['wp'], c['wff2'], c['bff2'], c['t'], c['wpo3'], c['bpo'], c['x_in'], out, B, ntok)


NAMES = ['P0(0)', 'P1(0)', 'P0(1)', 'P1(1)', 'FF(0)', 'P0(2)', 'P1(2)', 'FF(1)', 'P0(3)', 'P1(3)', 'FF(2)', 'FF(3)']
for abl in (20, 21, 22, 23, 10, 11):      # 10 a + x: ablation x with a loader waves
    dbg_fn(stamps.data_ptr(), abl)
    for _ in range(5):
        one()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        one()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    st = stamps.cpu().double()
    # stamps of wave 0: 0 entry, 1 prologue done, 2 ... 61 one per k-tile step (two units) of the twelve blocks, 62 FF-out epilogue done,
    # 63 ... 67 the proj_out k-tile steps (four units each), 68 ring drained, 69 end
    d = st[:, 1:70] - st[:, 0:69]
    mean = d.mean(0)
    tot = (st[:, 69] - st[:, 0])
    print(f'ABL {abl}: {us:7.1f} us per launch | workgroup cycles mean {tot.mean():9.0f} max {tot.max():9.0f} | prologue {mean[0]:6.0f} | '
          f'FF-out epilogue {mean[61]:6.0f} | proj_out {mean[62:67].sum() / 20:5.0f} cyc/unit | drain {mean[67]:6.0f} | final epilogue {mean[68]:6.0f}', flush=True)
    print('    cycles per unit by block: ' + '  '.join(f'{n} {mean[1 + 5 * b:6 + 5 * b].sum() / 10:4.0f}' for b, n in enumerate(NAMES)), flush=True)
    if abl % 10 == 0:
        print('    P1(1) per k-tile step (2 units):', ' '.join(f'{v:5.0f}' for v in mean[16:21]), '| FF(0):', ' '.join(f'{v:5.0f}' for v in mean[21:26]), flush=True)
        print(f'    per-workgroup total cycles: min {tot.min():.0f} p50 {tot.median():.0f} max {tot.max():.0f}', flush=True)
dbg_fn(None, 0)
