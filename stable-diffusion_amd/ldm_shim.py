"""The two thin wrappers between the samplers and the UNet, restated so the hot path can be driven without
pytorch_lightning: `LatentDiffusion.apply_model` (ldm/models/diffusion/ddpm.py:891-900,986-992) and
`DiffusionWrapper.forward` (ddpm.py:1402-1410, conditioning_key 'crossattn'), plus the schedule buffers
`DDPM.register_schedule` registers (ddpm.py:117-169) that the samplers read.

With the real `ldm` package installed the reference's own LatentDiffusion does this job;
this module exists for bench.py / tests / multi-GPU sampling where only the UNet path is needed.
"""
import numpy as np
import torch
import torch.nn as nn


def make_beta_schedule_linear(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    """util.py:21-25 with the SD-v1 yaml values (v1-inference.yaml:5-6,9)."""
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2


class DiffusionWrapperHIP(nn.Module):
    def __init__(self, diffusion_model):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.conditioning_key = 'crossattn'

    def forward(self, x, t, c_concat=None, c_crossattn=None):
        cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc)


class LatentDiffusionHIP(nn.Module):
    """What the samplers touch on `model`: num_timesteps, betas, alphas_cumprod(_prev), device, apply_model."""

    def __init__(self, unet, timesteps=1000, linear_start=0.00085, linear_end=0.0120):
        super().__init__()
        self.model = DiffusionWrapperHIP(unet)
        betas = make_beta_schedule_linear(timesteps, linear_start, linear_end)
        alphas_cumprod = np.cumprod(1. - betas, axis=0)
        self.num_timesteps = int(timesteps)
        self.parameterization = 'eps'
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.register_buffer('betas', f32(betas))
        self.register_buffer('alphas_cumprod', f32(alphas_cumprod))
        self.register_buffer('alphas_cumprod_prev', f32(np.append(1., alphas_cumprod[:-1])))
        self.register_buffer('sqrt_alphas_cumprod', f32(np.sqrt(alphas_cumprod)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', f32(np.sqrt(1. - alphas_cumprod)))

    @property
    def device(self):
        return self.betas.device

    def apply_model(self, x_noisy, t, cond, return_ids=False):
        if not isinstance(cond, dict):
            cond = {'c_crossattn': [cond] if not isinstance(cond, list) else cond}
        return self.model(x_noisy, t, **cond)

    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:274-277"""
        noise = torch.randn_like(x_start) if noise is None else noise
        sh = (t.shape[0],) + (1,) * (x_start.dim() - 1)
        return self.sqrt_alphas_cumprod.gather(-1, t).reshape(sh) * x_start + \
            self.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(sh) * noise
