"""`FrozenCLIPEmbedderHIP` -- drop-in for `ldm.modules.encoders.modules.FrozenCLIPEmbedder` on MI355X.

Plugged in through the reference's plugin mechanism (`instantiate_from_config(cond_stage_config)`,
ldm/models/diffusion/ddpm.py:509-520):

    cond_stage_config:
      target: stable_diffusion_amd.clip.FrozenCLIPEmbedderHIP

Same constructor (`version`, `device`, `max_length`; modules.py:139), same `forward(text)` / `encode(text)` returning
`last_hidden_state` [B, 77, 768], same parameter names (`transformer.text_model.*` as transformers 4.19.2 -- the version
the reference pins -- names them, so the `cond_stage_model.*` part of an SD checkpoint loads).  Tokenization stays on the
host (`CLIPTokenizer`, exactly the reference's call); the transformer runs in libsdmi.so.  No CPU / PyTorch fallback.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .unet import _Node

CLIP_VIT_L14_TEXT = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                         num_attention_heads=12, max_position_embeddings=77)


def make_clip_cfg(tc):
    if tc.get('hidden_act', 'quick_gelu') != 'quick_gelu':
        raise NotImplementedError('FrozenCLIPEmbedderHIP implements the quick_gelu CLIP text tower only')
    cfg = _lib.ClipCfg()
    cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size = int(tc['vocab_size']), int(tc['hidden_size']), int(tc['intermediate_size'])
    cfg.num_layers, cfg.num_heads, cfg.max_positions = int(tc['num_hidden_layers']), int(tc['num_attention_heads']), \
        int(tc['max_position_embeddings'])
    return cfg


class _ClipHandle:
    """Owns one sdmi_clip*."""

    def __init__(self, cfg):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.sdmi_clip_create(C.byref(cfg), C.byref(h)))
        self.h = h

    def weight_specs(self):
        n = self.lib.sdmi_clip_num_weights(self.h)
        out = []
        buf = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(n):
            _lib.check(self.lib.sdmi_clip_weight_info(self.h, i, buf, 256, shape, C.byref(nd)))
            out.append((buf.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def __del__(self):
        try:
            if self.h:
                self.lib.sdmi_clip_destroy(self.h)
                self.h = None
        except Exception:
            pass


class FrozenCLIPEmbedderHIP(nn.Module):
    MAX_BATCH = 64

    def __init__(self, version='openai/clip-vit-large-patch14', device='cuda', max_length=77, text_config=None,
                 tokenizer=None):
        super().__init__()
        if tokenizer is None:
            from transformers import CLIPTokenizer
            tokenizer = CLIPTokenizer.from_pretrained(version)         # modules.py:141
        self.tokenizer = tokenizer
        self.device = device
        self.max_length = max_length
        self.text_config = dict(CLIP_VIT_L14_TEXT if text_config is None else text_config)
        self._cfg = make_clip_cfg(self.text_config)
        self._handle = _ClipHandle(self._cfg)
        self._specs = self._handle.weight_specs()          # keys relative to `transformer.`
        self.add_module('transformer', _Node())
        for key, shape in self._specs:
            *path, leaf = key.split('.')
            node = self.transformer
            for name in path:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            node.register_parameter(leaf, nn.Parameter(torch.zeros(shape), requires_grad=False))
        # transformers 4.19.2 keeps position_ids as a persistent buffer: present in SD checkpoints
        self.transformer.text_model.embeddings.register_buffer(
            'position_ids', torch.arange(self._cfg.max_positions).expand((1, -1)).clone())
        self._packed_sig = None
        self._sentinels = None
        self._ws = None
        self.freeze()

    def freeze(self):
        """modules.py:146-149"""
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

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
            ps = dict(self.transformer.named_parameters())
            keys = [self._specs[0][0], self._specs[len(self._specs) // 2][0], self._specs[-1][0]]
            self._sentinels = [ps[k] for k in keys]
        return tuple((p.data_ptr(), p._version) for p in self._sentinels)

    def pack(self):
        lib = self._handle.lib
        stream = _lib.stream_ptr()
        ps = dict(self.transformer.named_parameters())
        for key, shape in self._specs:
            p = ps[key].detach()
            if not p.is_cuda:
                raise RuntimeError('FrozenCLIPEmbedderHIP parameters must live on the GPU (call model.cuda() first); '
                                   'there is no CPU implementation of this path')
            p = p.float().contiguous()
            shp = (C.c_int64 * len(shape))(*shape)
            _lib.check(lib.sdmi_clip_set_weight(self._handle.h, key.encode(), p.data_ptr(), shp, len(shape), stream))
        torch.cuda.current_stream().synchronize()
        _lib.check(lib.sdmi_clip_finalize(self._handle.h))
        self._sentinels = None
        self._packed_sig = self._signature()

    # ---- CLIPTextModel(input_ids).last_hidden_state ----------------------------------------------------------------------
    @torch.no_grad()
    def encode_ids(self, ids):
        if not ids.is_cuda:
            raise RuntimeError('FrozenCLIPEmbedderHIP runs on an MI355X device tensor only (no CPU fallback)')
        if ids.dim() != 2 or ids.shape[1] > self._cfg.max_positions:
            raise ValueError(f'input_ids must be [B, L <= {self._cfg.max_positions}]')
        if self._packed_sig is None or self._packed_sig != self._signature():
            self.pack()
        ids = ids.detach().to(torch.int64).contiguous()
        if int(ids.min()) < 0 or int(ids.max()) >= self._cfg.vocab_size:
            raise IndexError('token id out of range')           # what nn.Embedding raises in the reference
        B, L = ids.shape
        out = torch.empty((B, L, self._cfg.hidden_size), dtype=torch.float32, device=ids.device)
        for b0 in range(0, B, self.MAX_BATCH):
            nb = min(self.MAX_BATCH, B - b0)
            key = (nb, L, str(ids.device))
            if self._ws is None or self._ws[0] != key:
                need = self._handle.lib.sdmi_clip_workspace_bytes(self._handle.h, nb, L)
                if need <= 0:
                    _lib.check(-1)
                self._ws = (key, torch.empty(int(need), dtype=torch.uint8, device=ids.device))
            ws = self._ws[1]
            _lib.check(self._handle.lib.sdmi_clip_forward(self._handle.h, ids[b0:b0 + nb].data_ptr(), out[b0:b0 + nb].data_ptr(),
                                                          nb, L, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        return out

    def forward(self, text):
        """modules.py:150-160"""
        batch_encoding = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                                        return_overflowing_tokens=False, padding='max_length', return_tensors='pt')
        tokens = batch_encoding['input_ids'].to(self.device)
        return self.encode_ids(tokens)

    def encode(self, text):
        return self(text)
