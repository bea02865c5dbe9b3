#!/usr/bin/env python3
"""Repeat one committed UNet golden case N times in one process and print the error of every repetition (an intermittent
kernel bug shows as a varying error; a deterministic one as a constant).  GPU box only.

    python tools/unet_repeat.py --case sdv1_64x64 --reps 8
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--case', default='sdv1_64x64')
    ap.add_argument('--reps', type=int, default=8)
    a = ap.parse_args()
    from test_unet_gpu import CFGS, _model, make_inputs
    z = np.load(os.path.join(ROOT, 'tests', 'golden', f'unet_{a.case}.npz'))
    cfg_name = a.case.split('_')[0]
    cfg = CFGS[cfg_name]
    m, _ = _model(cfg_name, int(z['weight_seed']))
    x, t, ctx = make_inputs(cfg, int(z['batch']), int(z['h']), int(z['w']), seed=int(z['input_seed']), ctx_len=int(z['ctx_len']),
                            timesteps=tuple(int(v) for v in z['t']))
    ref = torch.from_numpy(z['eps'])
    xs, ts, cs = x.cuda(), t.cuda(), ctx.cuda()
    first = None
    worst = 0.0
    for r in range(a.reps):
        eps = m(xs, ts, context=cs).float().cpu()
        err = (eps - ref).abs()
        same = 'first' if first is None else ('same-bits' if torch.equal(eps, first) else f'differs {float((eps - first).abs().max()):.3e}')
        if first is None:
            first = eps
        worst = max(worst, float(err.max()))
        print(f'[{a.case} rep {r}] max-abs {float(err.max()):.3e} rms {float(err.pow(2).mean().sqrt()):.3e} vs rep 0: {same}', flush=True)
    return 1 if worst > 1e-3 else 0


if __name__ == '__main__':
    sys.exit(main())
