"""The composed txt2img path on the GPU vs the composed oracle (scripts/txt2img.py:298-315):

    token ids -> FrozenCLIPEmbedder (cond / uncond) -> PLMSSampler.sample (CFG) -> decode_first_stage (z / 0.18215,
    ddpm.py:713) -> clamp((x + 1) / 2, 0, 1)

Every stage has its own parity test; this one checks the composition (stage interfaces, the [uncond, cond] order, the
1 / scale_factor fold, the final clamp) on small configurations the CPU oracle finishes in seconds.  The sampler feeds
each eps back into the next step, so the per-call error (<= 1e-3) compounds: the tolerance is on the final image
(8-bit levels are 3.9e-3 apart after the (x + 1) / 2 map) and written below."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import clip_ref, samplers_ref, unet_ref, vae_ref  # noqa: E402
from oracle.plan import TINY  # noqa: E402
from oracle.weights import make_state_dict  # noqa: E402

# Budget: injecting the measured per-stage errors (eps rms 2e-4 / max 1e-3 per UNet call, context rms 6e-4) into the oracle
# pipeline moves the final latent by ~2.4e-2 (|z| max ~18 with these random weights at scale 7.5) and the image by 1.5e-3.
# Tolerances = 2x the errors measured on MI355X (round 4 / round 5 GPU logs; the per-call error does not compound over more steps here:
# the sampler contracts towards x0), per (sampler, steps): {image on [0, 1], final latent (|z| max ~18)}.
#   measured  plms S=5: 6.0e-4 ... 7.9e-4 / 7.6e-3     ddim S=5: 9.2e-4 ... 1.2e-3 / 1.3e-2
TOLS = {('plms', 5): (2.0e-3, 1.6e-2), ('ddim', 5): (2.5e-3, 2.6e-2), ('plms', 20): (2.5e-3, 2.6e-2), ('ddim', 20): (2.5e-3, 2.6e-2)}


def _text_config(cfg):
    return dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, max_position_embeddings=cfg.max_positions)


@pytest.mark.parametrize('sampler_name,S', [('plms', 5), ('ddim', 5), ('plms', 20), ('ddim', 20)])
def test_txt2img_pipeline_matches_oracle_pipeline(sampler_name, S):
    from stable_diffusion_amd import (AutoencoderKLHIP, DDIMSamplerHIP, FrozenCLIPEmbedderHIP, LatentDiffusionHIP,
                                      PLMSSamplerHIP, UNetModelHIP)
    ccfg, ucfg, vcfg = clip_ref.TINY_CLIP, TINY, vae_ref.TINY_VAE
    assert ccfg.hidden_size == ucfg.context_dim
    csd = clip_ref.make_clip_state_dict(ccfg, 0)
    usd = make_state_dict(ucfg, 0)
    vsd = vae_ref.make_vae_state_dict(vcfg, 0, encoder=False)
    L, scale, h, w = 77, 7.5, 16, 16
    ids = clip_ref.make_clip_ids(ccfg, 2, L, seed=3)              # row 0: the prompt, row 1: the "" of the uncond branch
    g = torch.Generator().manual_seed(42)
    x_T = torch.randn(1, 4, h, w, generator=g)

    # ---- oracle pipeline (CPU fp32) --------------------------------------------------------------------------------
    ctx = clip_ref.clip_text_forward(csd, ccfg, ids)
    c, uc = ctx[0:1], ctx[1:2]
    _, ac = samplers_ref.make_alphas_cumprod()
    apply_model = lambda x, t, cc: unet_ref.unet_forward(usd, ucfg, x, t, cc)
    fn = samplers_ref.plms_sample if sampler_name == 'plms' else samplers_ref.ddim_sample
    z_ref = fn(apply_model, ac, S, x_T, c, scale, uc)
    img_ref = torch.clamp((vae_ref.decode_first_stage(vsd, vcfg, z_ref) + 1.0) / 2.0, min=0.0, max=1.0)

    # ---- HIP pipeline (every stage through libsdmi) ------------------------------------------------------------------
    clip = FrozenCLIPEmbedderHIP(text_config=_text_config(ccfg), tokenizer=object())
    clip.load_state_dict({'transformer.' + k: v for k, v in csd.items()}, strict=False)
    clip = clip.cuda()
    unet = UNetModelHIP(**ucfg.ref_kwargs())
    unet.load_state_dict(usd, strict=True)
    ld = LatentDiffusionHIP(unet).cuda()
    vae = AutoencoderKLHIP(vcfg.ddconfig(), None, vcfg.embed_dim, parts=1)
    vae.load_state_dict(vsd, strict=True)
    vae = vae.cuda()
    ctx_h = clip.encode_ids(ids.cuda())
    c_h, uc_h = ctx_h[0:1], ctx_h[1:2]
    smp = (PLMSSamplerHIP if sampler_name == 'plms' else DDIMSamplerHIP)(ld)
    z_h, _ = smp.sample(S=S, batch_size=1, shape=[4, h, w], conditioning=c_h, verbose=False, x_T=x_T.cuda(),
                        unconditional_guidance_scale=scale, unconditional_conditioning=uc_h, eta=0.0)
    img_h = torch.clamp((vae.decode_first_stage(z_h) + 1.0) / 2.0, min=0.0, max=1.0)
    torch.cuda.synchronize()

    e_ctx = (ctx_h.float().cpu() - ctx).abs().max().item()
    e_z = (z_h.float().cpu() - z_ref).abs().max().item()
    e_img = (img_h.float().cpu() - img_ref).abs().max().item()
    print(f'[pipeline {sampler_name} S={S}] context err {e_ctx:.3e} | latent err {e_z:.3e} (|z| max {z_ref.abs().max():.2f}) | '
          f'image err {e_img:.3e} on [0,1] (mean {img_ref.mean():.3f}, frac clamped '
          f'{((img_ref == 0) | (img_ref == 1)).float().mean():.3f})', flush=True)
    assert img_h.shape == (1, 3, h * vae.factor, w * vae.factor) and torch.isfinite(img_h).all()
    IMG_TOL, LAT_TOL = TOLS[(sampler_name, S)]
    assert e_z <= LAT_TOL and e_img <= IMG_TOL, (e_z, LAT_TOL, e_img, IMG_TOL)
