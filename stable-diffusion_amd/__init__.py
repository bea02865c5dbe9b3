"""stable-diffusion_amd: MI355X-native txt2img sampling hot path (UNet eps prediction + PLMS/DDIM loop).

The arithmetic lives in libsdmi.so (hand-written gfx950 HIP kernels behind a C ABI, include/sdmi.h);
this package is the host-side mirror of the reference's interfaces for that path:

    UNetModelHIP      <- ldm.modules.diffusionmodules.openaimodel.UNetModel
    PLMSSamplerHIP    <- ldm.models.diffusion.plms.PLMSSampler
    DDIMSamplerHIP    <- ldm.models.diffusion.ddim.DDIMSampler
    DPMSolverSamplerHIP <- ldm.models.diffusion.dpm_solver.sampler.DPMSolverSampler
    AutoencoderKLHIP  <- ldm.models.autoencoder.AutoencoderKL (decode / encode)
    FrozenCLIPEmbedderHIP <- ldm.modules.encoders.modules.FrozenCLIPEmbedder

Importable as `stable_diffusion_amd` (see stable_diffusion_amd.py at the repo root).
"""
from . import _lib  # noqa: F401
from .unet import UNetModelHIP  # noqa: F401
from .samplers import PLMSSamplerHIP, DDIMSamplerHIP, DPMSolverSamplerHIP  # noqa: F401
from .vae import AutoencoderKLHIP  # noqa: F401
from .clip import FrozenCLIPEmbedderHIP  # noqa: F401
from .ldm_shim import LatentDiffusionHIP, DiffusionWrapperHIP  # noqa: F401
from . import debug, postprocess  # noqa: F401

__all__ = ['UNetModelHIP', 'AutoencoderKLHIP', 'FrozenCLIPEmbedderHIP', 'PLMSSamplerHIP', 'DDIMSamplerHIP', 'DPMSolverSamplerHIP', 'LatentDiffusionHIP', 'DiffusionWrapperHIP']
