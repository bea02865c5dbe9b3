"""CPU restatement (fp32, functional torch) of the text encoder -- SURVEY.md 8(f-2).

The reference's `FrozenCLIPEmbedder.forward` (ldm/modules/encoders/modules.py:150-160) tokenizes, then returns
`self.transformer(input_ids=tokens).last_hidden_state`, where `self.transformer` is Hugging Face `CLIPTextModel`
(`transformers==4.19.2`, environment.yaml:25 -- a third-party dependency, absent from /root/reference).  This file
restates that model's published algorithm (transformers/models/clip/modeling_clip.py: CLIPTextEmbeddings,
CLIPEncoderLayer, CLIPAttention with the causal mask of CLIPTextTransformer, CLIPMLP with quick_gelu, final_layer_norm):

    x = token_embedding[ids] + position_embedding[0..L-1]
    per layer:  h = LN1(x); q, k, v = Linear(h) (+bias), split into heads; att = softmax(q k^T / sqrt(d) + causal) v
                x = x + out_proj(att);   x = x + fc2(quick_gelu(fc1(LN2(x))))          quick_gelu(u) = u * sigmoid(1.702 u)
    last_hidden_state = final_layer_norm(x)

Pinning: `oracle/make_golden_clip.py` loads `make_clip_state_dict` into the `CLIPTextModel` of the transformers version
installed in the build container (5.x: same arithmetic; its state_dict drops the `text_model.` prefix, mapped there),
asserts this restatement equals it and freezes its outputs as tests/golden/clip_*.npz.  Key names here are the
4.19.2 ones (`text_model.embeddings...`), i.e. the `cond_stage_model.transformer.` sub-tree of an SD checkpoint.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product path (stable-diffusion_amd/) never imports this.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class CLIPTextCfg:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_layers: int = 12
    num_heads: int = 12
    max_positions: int = 77


SD_CLIP = CLIPTextCfg()                                     # openai/clip-vit-large-patch14 text tower
TINY_CLIP = CLIPTextCfg(vocab_size=1000, hidden_size=128, intermediate_size=512, num_layers=2, num_heads=2)


def clip_param_specs(cfg: CLIPTextCfg):
    C, I = cfg.hidden_size, cfg.intermediate_size
    specs = [('text_model.embeddings.token_embedding.weight', (cfg.vocab_size, C), 'emb'),
             ('text_model.embeddings.position_embedding.weight', (cfg.max_positions, C), 'emb')]
    for i in range(cfg.num_layers):
        p = f'text_model.encoder.layers.{i}.'
        for n in ('q_proj', 'k_proj', 'v_proj'):
            specs += [(p + f'self_attn.{n}.weight', (C, C), 'w'), (p + f'self_attn.{n}.bias', (C,), 'b')]
        specs += [(p + 'self_attn.out_proj.weight', (C, C), 'w'), (p + 'self_attn.out_proj.bias', (C,), 'b'),
                  (p + 'layer_norm1.weight', (C,), 'gamma'), (p + 'layer_norm1.bias', (C,), 'beta'),
                  (p + 'mlp.fc1.weight', (I, C), 'w'), (p + 'mlp.fc1.bias', (I,), 'b'),
                  (p + 'mlp.fc2.weight', (C, I), 'w'), (p + 'mlp.fc2.bias', (C,), 'b'),
                  (p + 'layer_norm2.weight', (C,), 'gamma'), (p + 'layer_norm2.bias', (C,), 'beta')]
    specs += [('text_model.final_layer_norm.weight', (C,), 'gamma'), ('text_model.final_layer_norm.bias', (C,), 'beta')]
    return specs


def make_clip_state_dict(cfg: CLIPTextCfg, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for key, shape, kind in clip_param_specs(cfg):
        if kind == 'w':
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        elif kind == 'emb':
            t = torch.randn(shape, generator=g) * 0.5
        elif kind == 'b':
            t = torch.randn(shape, generator=g) * 0.05
        elif kind == 'gamma':
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        sd[key] = t.float()
    return sd


def make_clip_ids(cfg: CLIPTextCfg, batch, L, seed=1):
    """token ids shaped like the tokenizer's output: BOS, words, EOS, EOS padding (values only matter as indices)"""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, cfg.vocab_size - 2, (batch, L), generator=g)
    ids[:, 0] = cfg.vocab_size - 2
    for b in range(batch):
        n = 3 + (7 * b + 5) % (L - 3)
        ids[b, n:] = cfg.vocab_size - 1
    return ids


@torch.no_grad()
def clip_text_forward(sd, cfg: CLIPTextCfg, ids):
    B, L = ids.shape
    C, H = cfg.hidden_size, cfg.num_heads
    d = C // H
    x = sd['text_model.embeddings.token_embedding.weight'][ids] + sd['text_model.embeddings.position_embedding.weight'][:L][None]
    mask = torch.full((L, L), float('-inf')).triu(1)            # query i sees keys <= i
    for i in range(cfg.num_layers):
        p = f'text_model.encoder.layers.{i}.'
        h = F.layer_norm(x, (C,), sd[p + 'layer_norm1.weight'], sd[p + 'layer_norm1.bias'], 1e-5)
        q, k, v = (F.linear(h, sd[p + f'self_attn.{n}.weight'], sd[p + f'self_attn.{n}.bias'])
                   .view(B, L, H, d).transpose(1, 2) for n in ('q_proj', 'k_proj', 'v_proj'))
        att = torch.softmax(q @ k.transpose(-1, -2) * (d ** -0.5) + mask, dim=-1) @ v
        att = att.transpose(1, 2).reshape(B, L, C)
        x = x + F.linear(att, sd[p + 'self_attn.out_proj.weight'], sd[p + 'self_attn.out_proj.bias'])
        h = F.layer_norm(x, (C,), sd[p + 'layer_norm2.weight'], sd[p + 'layer_norm2.bias'], 1e-5)
        h = F.linear(h, sd[p + 'mlp.fc1.weight'], sd[p + 'mlp.fc1.bias'])
        h = h * torch.sigmoid(1.702 * h)
        x = x + F.linear(h, sd[p + 'mlp.fc2.weight'], sd[p + 'mlp.fc2.bias'])
    return F.layer_norm(x, (C,), sd['text_model.final_layer_norm.weight'], sd['text_model.final_layer_norm.bias'], 1e-5)
