"""Parity on REAL weights, the day a checkpoint is present:  SD_CKPT=/path/to/sd-v1-4.ckpt python -m pytest tests -m gpu -k real_checkpoint

There is no Stable Diffusion checkpoint in the build or GPU environments (no network), so every parity number of this repository is
on seeded random weights in the exact SD-v1 architecture.  This test is skipped until `SD_CKPT` names a checkpoint
(`scripts/txt2img.py:49-66`: `torch.load(ckpt)["state_dict"]`, keys `model.diffusion_model.*`); then it holds the UNet to the same
bar (north_star: eps max-abs <= 1e-3 vs the fp32 reference on identical (x_t, t, context)) at 16 x 16 and 64 x 64 latents, with the
fp16 range guard on (csrc/range.hip: activations beyond the fp16 range would be reported instead of silently saturating).
Reference = the CPU oracle (oracle/unet_ref.py, pinned to the reference UNetModel at 0.0 difference by oracle/make_golden.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
CKPT = os.environ.get('SD_CKPT', '')
TOL = 1e-3


@pytest.mark.skipif(not (CKPT and os.path.exists(CKPT)), reason='SD_CKPT does not name a Stable Diffusion v1 checkpoint')
@pytest.mark.parametrize('h,w', [(16, 16), (64, 64)])
def test_real_checkpoint_unet_eps_matches_the_fp32_reference(h, w):
    from oracle import unet_ref
    from oracle.plan import SD_V1
    from oracle.weights import make_inputs
    from stable_diffusion_amd import UNetModelHIP
    from stable_diffusion_amd.debug import range_check, range_report
    blob = torch.load(CKPT, map_location='cpu')
    sd_all = blob.get('state_dict', blob)
    pre = 'model.diffusion_model.'
    sd = {k[len(pre):]: v.float() for k, v in sd_all.items() if k.startswith(pre)}
    assert len(sd) == 686, f'{len(sd)} UNet tensors under {pre!r}: not an SD-v1 checkpoint?'
    m = UNetModelHIP(**SD_V1.ref_kwargs())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x, t, ctx = make_inputs(SD_V1, 2, h, w, seed=11, ctx_len=77, timesteps=(981, 481))
    range_check(True)
    try:
        eps = m(x.cuda(), t.cuda(), context=ctx.cuda())
        torch.cuda.synchronize()
        rep = range_report()
    finally:
        range_check(False)
    assert rep['over_6e4'] == 0 and rep['nonfinite'] == 0, f'activations left the fp16 range: {rep}'
    ref = unet_ref.unet_forward(sd, SD_V1, x, t, ctx)
    err = (eps.float().cpu() - ref).abs()
    print(f'[real checkpoint {h}x{w}] max-abs {err.max():.3e} rms {err.pow(2).mean().sqrt():.3e} |eps|max {ref.abs().max():.3f}', flush=True)
    assert float(err.max()) <= TOL
