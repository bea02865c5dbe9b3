"""First-stage (AutoencoderKL) parity on the GPU: AutoencoderKLHIP (through the C ABI) vs the committed reference goldens.

The goldens are outputs of the *reference* `Decoder` / `Encoder` (+ the two 1x1 quant convs, fp32, CPU) produced by
oracle/make_golden_vae.py; weights and inputs are regenerated here from the same seeds (oracle.vae_ref), so nothing
under /root/reference is needed.

Tolerances (written here, measured values are printed): the decoded image lives in about [-3.5, 3.5] for these synthetic
weights (rms 0.6); the path rounds every conv operand to fp16 once (2^-11 relative) and accumulates in fp32, which gives
~1e-3 max-abs over 0.8 M output values after ~30 convs.  DEC_TOL = 4e-3 is half of one 8-bit image level
(1/255 of the [-1, 1] range = 7.8e-3).  Moments of the encoder: same budget."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.vae_ref import SD_VAE, SMALL_VAE, TINY_VAE, make_vae_inputs, make_vae_state_dict  # noqa: E402

# Round 6: 1.25 x the worst error MEASURED over weight seeds 0 (the reference goldens), 1 and 2 (against the oracle, which the
# golden script pins to the reference at 0.0) -- rounds 1-5 carried 4e-3 for both.  Measured worst (pass H / I, MI355X): decode 2.34e-3
# (sd_64x64, seed 0), encode moments 2.77e-3 (sd_256x256, seed 0); seeds 1 / 2 (test_vae_other_weight_seeds): decode <= 1.90e-3, encode <= 2.07e-3.
DEC_TOL = 3.0e-3
ENC_TOL = 3.5e-3
CFGS = {'tiny': TINY_VAE, 'small': SMALL_VAE, 'sd': SD_VAE}
_models = {}


def _model(cfg_name, wseed):
    key = (cfg_name, wseed)
    if key not in _models:
        _models.clear()
        torch.cuda.empty_cache()
        from stable_diffusion_amd import AutoencoderKLHIP
        cfg = CFGS[cfg_name]
        sd = make_vae_state_dict(cfg, wseed)
        m = AutoencoderKLHIP(cfg.ddconfig(), {'target': 'torch.nn.Identity'}, cfg.embed_dim)
        m.load_state_dict(sd, strict=True)          # same keys / shapes as the reference state_dict
        _models[key] = m.cuda().eval()
    return _models[key]


DEC_CASES = ['tiny_8x8', 'tiny_8x24', 'small_16x16', 'sd_8x8', 'sd_16x24', 'sd_32x32', 'sd_64x64']
ENC_CASES = ['tiny_32x32', 'tiny_16x48', 'sd_64x64', 'sd_128x192', 'sd_256x256']


@pytest.mark.parametrize('case', DEC_CASES)
def test_vae_decode_matches_reference_golden(case, golden_dir):
    z = np.load(os.path.join(golden_dir, f'vae_dec_{case}.npz'))
    cfg_name = str(z['cfg'])
    cfg = CFGS[cfg_name]
    m = _model(cfg_name, int(z['weight_seed']))
    lat = make_vae_inputs(cfg, int(z['batch']), int(z['h']), int(z['w']), seed=int(z['input_seed']))
    img = m.decode(lat.cuda())
    torch.cuda.synchronize()
    ref = torch.from_numpy(z['out'])
    err = (img.float().cpu() - ref).abs()
    print(f'[vae decode {case}] HIP-vs-reference(fp32) max-abs {err.max():.3e} rms {err.pow(2).mean().sqrt():.3e} '
          f'|img|max {ref.abs().max():.3f} nan={bool(torch.isnan(img).any())}', flush=True)
    assert img.shape == ref.shape and img.dtype == torch.float32
    assert float(err.max()) <= DEC_TOL


@pytest.mark.parametrize('case', ENC_CASES)
def test_vae_encode_matches_reference_golden(case, golden_dir):
    z = np.load(os.path.join(golden_dir, f'vae_enc_{case}.npz'))
    cfg_name = str(z['cfg'])
    cfg = CFGS[cfg_name]
    m = _model(cfg_name, int(z['weight_seed']))
    g = torch.Generator().manual_seed(int(z['input_seed']))
    x = torch.rand(int(z['batch']), cfg.in_channels, int(z['h']), int(z['w']), generator=g) * 2 - 1
    mom = m.encode_moments(x.cuda())
    torch.cuda.synchronize()
    ref = torch.from_numpy(z['moments'])
    err = (mom.float().cpu() - ref).abs()
    print(f'[vae encode {case}] HIP-vs-reference(fp32) max-abs {err.max():.3e} rms {err.pow(2).mean().sqrt():.3e} '
          f'|moments|max {ref.abs().max():.3f} nan={bool(torch.isnan(mom).any())}', flush=True)
    assert mom.shape == ref.shape
    assert float(err.max()) <= ENC_TOL
    post = m.encode(x.cuda())
    assert torch.equal(post.mode(), mom[:, :cfg.embed_dim])
    assert torch.allclose(post.std, torch.exp(0.5 * mom[:, cfg.embed_dim:].clamp(-30.0, 20.0)))


@pytest.mark.parametrize('wseed', [1, 2])
def test_vae_other_weight_seeds(wseed):
    """The SD first stage under weight seeds the goldens do not use, against the oracle on the same inputs (oracle == reference Encoder / Decoder
    at 0.0 on every golden: oracle/make_golden_vae.py): decode 16x24 and 32x32 latents, encode a 64x64 image."""
    from oracle import vae_ref
    cfg = SD_VAE
    m = _model('sd', wseed)
    sd = make_vae_state_dict(cfg, wseed)
    for h, w in ((16, 24), (32, 32)):
        lat = make_vae_inputs(cfg, 1, h, w, seed=3)
        ref = vae_ref.vae_decode(sd, cfg, lat)
        err = (m.decode(lat.cuda()).float().cpu() - ref).abs()
        print(f'[vae decode sd_{h}x{w} weight seed {wseed}] HIP-vs-oracle(fp32) max-abs {err.max():.3e} rms {err.pow(2).mean().sqrt():.3e}', flush=True)
        assert float(err.max()) <= DEC_TOL
    g = torch.Generator().manual_seed(4)
    x = torch.rand(1, cfg.in_channels, 64, 64, generator=g) * 2 - 1
    ref = vae_ref.vae_encode_moments(sd, cfg, x)
    err = (m.encode_moments(x.cuda()).float().cpu() - ref).abs()
    print(f'[vae encode sd_64x64 weight seed {wseed}] HIP-vs-oracle(fp32) max-abs {err.max():.3e} rms {err.pow(2).mean().sqrt():.3e}', flush=True)
    assert float(err.max()) <= ENC_TOL


def test_vae_decode_is_repeatable_and_scaled():
    """bit-identical run to run (no atomics on floats anywhere), and decode_first_stage's 1/scale_factor fold equals
    scaling the latent first (ddpm.py:713)."""
    cfg = SD_VAE
    m = _model('sd', 0)
    lat = make_vae_inputs(cfg, 2, 16, 16, seed=5).cuda()
    a = m.decode(lat)
    b = m.decode(lat)
    assert torch.equal(a, b)
    c = m.decode_first_stage(lat * 0.18215)
    # (z * s) * (1/s) differs from z by one fp32 rounding of the latent; that flips a few fp16 operand roundings
    # downstream, so the two images agree to the parity tolerance, not bit for bit
    assert (a - c).abs().max().item() < DEC_TOL


def test_vae_batches_are_independent():
    """B images decoded together == decoded one by one (the reference has no cross-sample op; GN statistics are per sample)."""
    m = _model('sd', 0)
    lat = make_vae_inputs(SD_VAE, 3, 8, 16, seed=9).cuda()
    all3 = m.decode(lat)
    for i in range(3):
        one = m.decode(lat[i:i + 1])
        assert (one - all3[i:i + 1]).abs().max().item() < 2e-3     # split-K / tile choice may differ with M


def test_vae_round_trip_shapes_and_refusals():
    m = _model('tiny', 0)
    x = torch.rand(1, 3, 32, 48, device='cuda') * 2 - 1
    rec, post = m(x, sample_posterior=False)
    assert rec.shape == x.shape and post.mean.shape == (1, 4, 16, 24)
    with pytest.raises(ValueError):
        m.encode_moments(torch.zeros(1, 3, 33, 48, device='cuda'))
    with pytest.raises(RuntimeError, match='no CPU'):
        m.decode(torch.zeros(1, 4, 8, 8))


def test_vae_decode_outputs_beyond_2_gb():
    """Four 768 x 768 images decoded together: the upsampling convs' fp32 outputs are 4 x 604 MB = 2.4 GB, i.e. byte offsets beyond
    2^31 -- where the write-through (buffer) stores of the GEMM epilogues, the split-K reduce and GroupNorm-apply must take their
    64-bit form (csrc/common.h sdmi_st_wt16: a 32-bit offset past the 2^31 records would be dropped by the range check, one past
    2^32 would wrap onto an earlier sample).  Every sample, the last one in particular, must equal its own single-image decode."""
    m = _model('sd', 0)
    free, _ = torch.cuda.mem_get_info()
    if free < 60 << 30:
        pytest.skip('needs ~40 GB of free HBM')
    lat = make_vae_inputs(SD_VAE, 4, 96, 96, seed=21).cuda()
    all4 = m.decode(lat)
    torch.cuda.synchronize()
    assert torch.isfinite(all4).all()
    for i in (0, 3):
        one = m.decode(lat[i:i + 1])
        err = (one - all4[i:i + 1]).abs().max().item()
        print(f'[vae decode 4 x 768x768] sample {i} vs its single-image decode: max-abs {err:.3e}', flush=True)
        assert err < 2e-3
    del all4
    torch.cuda.empty_cache()
