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


for abl in (0, 1, 2, 3, 4):
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
    # stamp order: 0 entry, 1 prologue done, then per chunk h: 5 + 5 GEGLU k-tile steps, 5 FF-out k-tile steps (15 per chunk),
    # then FF-out epilogue start (62), 5 proj_out k-tile stamps (63..67), drained (68), end (69)
    d = st[:, 1:70] - st[:, 0:69]
    mean = d.mean(0)
    tot = (st[:, 69] - st[:, 0])
    line = f'ABL {abl}: {us:7.1f} us per launch | workgroup cycles mean {tot.mean():9.0f} max {tot.max():9.0f} | prologue {mean[0]:7.0f}'
    for h in range(4):
        b = 1 + 15 * h
        line += f' | h{h} geglu {mean[b:b + 10].sum() / 20:6.0f} ff {mean[b + 10:b + 15].sum() / 10:6.0f} cyc/unit'
    line += f' | ff-epi {mean[61] if False else (st[:, 62] - st[:, 61]).mean():6.0f}'
    line += f' | proj_out {(st[:, 67] - st[:, 62]).mean() / 20:6.0f} cyc/unit (incl. ff-out epilogue) | drain {(st[:, 68] - st[:, 67]).mean():6.0f} | epilogue {(st[:, 69] - st[:, 68]).mean():6.0f}'
    print(line, flush=True)
    if abl == 0:
        # first k-tile step of each GEGLU pass carries the previous epilogue: print the ten steps of chunk 1
        print('   chunk 1 per k-tile step (2 units):', ' '.join(f'{v:5.0f}' for v in mean[16:31]), flush=True)
        per_wg = tot
        print(f'   per-workgroup total cycles: min {per_wg.min():.0f} p50 {per_wg.median():.0f} max {per_wg.max():.0f}; '
              f'start skew (stamp 0 spread) {st[:, 0].max() - st[:, 0].min():.0f} cycles', flush=True)
dbg_fn(None, 0)
