"""`AutoencoderKLHIP` -- drop-in for `ldm.models.autoencoder.AutoencoderKL` (inference only) on MI355X.

Plugged in through the reference's plugin mechanism (`instantiate_from_config`, ldm/util.py:78-93;
`LatentDiffusion.instantiate_first_stage`, ldm/models/diffusion/ddpm.py:502-507): a copy of
configs/stable-diffusion/v1-inference.yaml with

    first_stage_config:
      target: stable_diffusion_amd.vae.AutoencoderKLHIP

Same constructor keywords (autoencoder.py:286-295), same parameter names (`encoder.*`, `decoder.*`, `quant_conv.*`,
`post_quant_conv.*`, so the `first_stage_model.*` part of an SD checkpoint loads), same calls:
`decode(z)` (autoencoder.py:330-333, reached from `decode_first_stage`, ddpm.py:705-763) and `encode(x)`
(autoencoder.py:324-328, reached from `encode_first_stage`, ddpm.py:825-863) which returns the reference's
`DiagonalGaussianDistribution` when `ldm` is importable (`get_first_stage_encoding` type-checks it, ddpm.py:542-549).
The arithmetic runs in libsdmi.so; there is no CPU / PyTorch fallback.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .unet import _Node


class DiagonalGaussianDistributionHIP:
    """ldm/modules/distributions/distributions.py:24-63 (the members the sampling scripts use)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean


def _posterior(moments):
    try:        # inside the reference's LatentDiffusion the posterior must be the reference's own class
        from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
        return DiagonalGaussianDistribution(moments)
    except ImportError:
        return DiagonalGaussianDistributionHIP(moments)


def make_vae_cfg(ddconfig, embed_dim):
    dd = dict(ddconfig)
    unsupported = []
    if list(dd.get('attn_resolutions', [])): unsupported.append('attn_resolutions=[] (mid-block attention only)')
    if dd.get('dropout', 0.0) != 0: unsupported.append('dropout=0')
    if not dd.get('double_z', True): unsupported.append('double_z=True')
    if not dd.get('resamp_with_conv', True): unsupported.append('resamp_with_conv=True')
    if dd.get('use_linear_attn', False) or dd.get('attn_type', 'vanilla') != 'vanilla': unsupported.append("attn_type='vanilla'")
    if dd.get('tanh_out', False) or dd.get('give_pre_end', False): unsupported.append('tanh_out=False, give_pre_end=False')
    if unsupported:
        raise NotImplementedError('AutoencoderKLHIP supports the SD-v1 first stage family only; needs: ' + '; '.join(unsupported))
    cfg = _lib.VaeCfg()
    ch_mult = list(dd.get('ch_mult', (1, 2, 4, 8)))
    if len(ch_mult) > 8:
        raise ValueError('at most 8 levels')
    cfg.ch, cfg.out_ch, cfg.n_levels = int(dd['ch']), int(dd['out_ch']), len(ch_mult)
    for i, m in enumerate(ch_mult):
        cfg.ch_mult[i] = int(m)
    cfg.num_res_blocks, cfg.in_channels = int(dd['num_res_blocks']), int(dd['in_channels'])
    cfg.z_channels, cfg.embed_dim = int(dd['z_channels']), int(embed_dim)
    return cfg


class _VaeHandle:
    """Owns one sdmi_vae*."""

    def __init__(self, cfg, parts):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.sdmi_vae_create(C.byref(cfg), parts, C.byref(h)))
        self.h = h

    def weight_specs(self):
        n = self.lib.sdmi_vae_num_weights(self.h)
        out = []
        buf = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(n):
            _lib.check(self.lib.sdmi_vae_weight_info(self.h, i, buf, 256, shape, C.byref(nd)))
            out.append((buf.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def __del__(self):
        try:
            if self.h:
                self.lib.sdmi_vae_destroy(self.h)
                self.h = None
        except Exception:
            pass


class AutoencoderKLHIP(nn.Module):
    MAX_BATCH = 8     # images per library call (larger batches are looped)

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key='image',
                 colorize_nlabels=None, monitor=None, parts=3):
        super().__init__()
        self.image_key = image_key
        self.embed_dim = int(embed_dim)
        self.ddconfig = dict(ddconfig)
        self._cfg = make_vae_cfg(ddconfig, embed_dim)
        self._parts = parts
        self._handle = _VaeHandle(self._cfg, parts)
        self._specs = self._handle.weight_specs()
        for key, shape in self._specs:
            *path, leaf = key.split('.')
            node = self
            for name in path:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            node.register_parameter(leaf, nn.Parameter(torch.zeros(shape), requires_grad=False))
        if monitor is not None:
            self.monitor = monitor
        self._packed_sig = None
        self._sentinels = None
        self._ws = {}
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    @property
    def factor(self):
        return 2 ** (self._cfg.n_levels - 1)

    def init_from_ckpt(self, path, ignore_keys=list()):
        """autoencoder.py:312-321"""
        sd = torch.load(path, map_location='cpu')['state_dict']
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        self.load_state_dict(sd, strict=False)

    # ---- weights -> library (same dirty tracking as UNetModelHIP) --------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._packed_sig = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed_sig = None
        return super().load_state_dict(*a, **k)

    def mark_dirty(self):
        self._packed_sig = None

    def _signature(self):
        if getattr(self, '_sentinels', None) is None:
            ps = dict(self.named_parameters())
            keys = [self._specs[0][0], self._specs[len(self._specs) // 2][0], self._specs[-1][0]]
            self._sentinels = [ps[k] for k in keys]
        return tuple((p.data_ptr(), p._version) for p in self._sentinels)

    def pack(self):
        lib = self._handle.lib
        stream = _lib.stream_ptr()
        ps = dict(self.named_parameters())
        for key, shape in self._specs:
            p = ps[key].detach()
            if not p.is_cuda:
                raise RuntimeError('AutoencoderKLHIP parameters must live on the GPU (call model.cuda() first); '
                                   'there is no CPU implementation of this path')
            p = p.float().contiguous()
            shp = (C.c_int64 * len(shape))(*shape)
            _lib.check(lib.sdmi_vae_set_weight(self._handle.h, key.encode(), p.data_ptr(), shp, len(shape), stream))
        torch.cuda.current_stream().synchronize()
        _lib.check(lib.sdmi_vae_finalize(self._handle.h))
        self._sentinels = None
        self._packed_sig = self._signature()

    def _ready(self, x):
        if not x.is_cuda:
            raise RuntimeError('AutoencoderKLHIP runs on an MI355X device tensor only (no CPU fallback)')
        if self._packed_sig is None or self._packed_sig != self._signature():
            self.pack()

    def _workspace(self, kind, B, H, W, device):
        key = (kind, B, H, W, str(device))
        if key not in self._ws:
            fn = self._handle.lib.sdmi_vae_decode_workspace_bytes if kind == 'dec' else \
                self._handle.lib.sdmi_vae_encode_workspace_bytes
            need = fn(self._handle.h, B, H, W)
            if need <= 0:
                _lib.check(-1)
            for k in [k for k in self._ws if k[0] == kind]:     # one live workspace per direction (img2img alternates
                del self._ws[k]                                  # encode and decode for every image)
            self._ws[key] = torch.empty(int(need), dtype=torch.uint8, device=device)
        return self._ws[key]

    # ---- AutoencoderKL.decode (autoencoder.py:330-333) ---------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z, z_scale=1.0):
        """z [B, embed_dim, h, w] -> image [B, out_ch, h*f, w*f] fp32.  `z_scale` multiplies z first (the 1/scale_factor
        of decode_first_stage, ddpm.py:713, folded into the first kernel)."""
        self._ready(z)
        B, Cz, H, W = z.shape
        assert Cz == self.embed_dim
        z32 = z.detach().float().contiguous()
        f = self.factor
        out = torch.empty((B, self._cfg.out_ch, H * f, W * f), dtype=torch.float32, device=z.device)
        for b0 in range(0, B, self.MAX_BATCH):
            nb = min(self.MAX_BATCH, B - b0)
            ws = self._workspace('dec', nb, H, W, z.device)
            _lib.check(self._handle.lib.sdmi_vae_decode(self._handle.h, z32[b0:b0 + nb].data_ptr(), float(z_scale),
                                                        out[b0:b0 + nb].data_ptr(), nb, H, W, ws.data_ptr(), ws.numel(),
                                                        _lib.stream_ptr()))
        return out

    # ---- AutoencoderKL.encode (autoencoder.py:324-328) ---------------------------------------------------------------
    @torch.no_grad()
    def encode_moments(self, x):
        self._ready(x)
        B, Cin, H, W = x.shape
        assert Cin == self._cfg.in_channels
        f = self.factor
        if H % f or W % f:
            raise ValueError(f'image sides must be multiples of {f}')
        x32 = x.detach().float().contiguous()
        out = torch.empty((B, 2 * self.embed_dim, H // f, W // f), dtype=torch.float32, device=x.device)
        for b0 in range(0, B, self.MAX_BATCH):
            nb = min(self.MAX_BATCH, B - b0)
            ws = self._workspace('enc', nb, H, W, x.device)
            _lib.check(self._handle.lib.sdmi_vae_encode(self._handle.h, x32[b0:b0 + nb].data_ptr(),
                                                        out[b0:b0 + nb].data_ptr(), nb, H, W, ws.data_ptr(), ws.numel(),
                                                        _lib.stream_ptr()))
        return out

    def encode(self, x):
        return _posterior(self.encode_moments(x))

    def forward(self, input, sample_posterior=True):
        """autoencoder.py:335-342"""
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior

    # ---- LatentDiffusion.decode_first_stage / encode_first_stage for callers without the reference's LatentDiffusion ---
    def decode_first_stage(self, z, scale_factor=0.18215):
        return self.decode(z, z_scale=1.0 / scale_factor)

    def encode_first_stage(self, x, scale_factor=0.18215, sample=True):
        p = self.encode(x)
        return scale_factor * (p.sample() if sample else p.mode())
