"""Seeded random weights in the SD-v1 UNet architecture (there is no checkpoint in the build environment).

Used by bench.py, the launcher and smoke runs.  Default-initialised reference weights give eps == 0
(`zero_module`, openaimodel.py:229-231,685; attention.py:244-248), so every tensor is drawn here.
Throughput is value independent; the values only need to keep activations O(1).
"""
import math

import torch


@torch.no_grad()
def randomize_(unet, seed=0):
    """In-place init of a UNetModelHIP (or anything with the same parameter names)."""
    dev = next(unet.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in unet.named_parameters():
        if name.endswith('.weight') and p.dim() >= 2:
            fan_in = p[0].numel()
            zero_init = name.endswith('out_layers.3.weight') or name.endswith('proj_out.weight') or name == 'out.2.weight'
            std = (0.5 if zero_init else 0.577) / math.sqrt(fan_in)
            p.copy_(torch.randn(p.shape, generator=g, device=dev) * std)
        elif name.endswith('.weight'):      # norm gamma
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=dev))
        else:                               # biases / norm beta
            p.copy_(0.02 * torch.randn(p.shape, generator=g, device=dev))
    if hasattr(unet, 'mark_dirty'):
        unet.mark_dirty()
    return unet


@torch.no_grad()
def randomize_vae_(vae, seed=0):
    """In-place unit-gain init of an AutoencoderKLHIP (or anything with the same parameter names)."""
    dev = next(vae.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in vae.named_parameters():
        if name.endswith('.weight') and p.dim() >= 2:
            p.copy_(torch.randn(p.shape, generator=g, device=dev) / math.sqrt(p[0].numel()))
        elif name.endswith('.weight'):
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=dev))
        else:
            p.copy_(0.02 * torch.randn(p.shape, generator=g, device=dev))
    if hasattr(vae, 'mark_dirty'):
        vae.mark_dirty()
    return vae


def synthetic_state_dict(unet_kwargs, seed=0):
    """CPU state_dict (reference key names) of a seeded random SD-v1-architecture UNet."""
    from .unet import UNetModelHIP
    m = UNetModelHIP(**unet_kwargs)
    randomize_(m, seed)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def synthetic_vae_state_dict(ddconfig, embed_dim=4, seed=0):
    """CPU state_dict (reference key names: encoder.*, decoder.*, quant_conv.*, post_quant_conv.*) of a seeded random
    first stage in the given architecture."""
    from .vae import AutoencoderKLHIP
    m = AutoencoderKLHIP(ddconfig, None, embed_dim)
    randomize_vae_(m, seed)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


@torch.no_grad()
def synthetic_clip_state_dict(text_config=None, seed=0):
    """CPU state_dict (keys `transformer.text_model.*`, as under `cond_stage_model.` in an SD checkpoint) of a seeded
    random CLIP text tower."""
    from .clip import FrozenCLIPEmbedderHIP
    m = FrozenCLIPEmbedderHIP(text_config=text_config, tokenizer=object())
    g = torch.Generator().manual_seed(seed)
    for name, p in m.named_parameters():
        if 'embedding' in name:
            p.copy_(torch.randn(p.shape, generator=g) * 0.5)
        elif name.endswith('.weight') and p.dim() == 2:
            p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p.shape[1]))
        elif name.endswith('.weight'):
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
        else:
            p.copy_(0.02 * torch.randn(p.shape, generator=g))
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


SD_V1_UNET_KWARGS = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320,
                         attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8,
                         use_spatial_transformer=True, transformer_depth=1, context_dim=768, use_checkpoint=True,
                         legacy=False)   # configs/stable-diffusion/v1-inference.yaml:29-44

SD_V1_VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                          ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)   # yaml:51-65
