"""Text-encoder parity on the GPU: FrozenCLIPEmbedderHIP (through the C ABI) vs goldens produced by Hugging Face
`CLIPTextModel` itself (oracle/make_golden_clip.py); weights / token ids are regenerated here from the same seeds.

Tolerance: last_hidden_state is LayerNorm output (rms 1, |x| max ~4.3); every GEMM operand is rounded to fp16 once with
fp32 accumulation, 12 layers -> measured max-abs is printed; CLIP_TOL = 4e-3 (0.1 % of the value range)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import clip_ref  # noqa: E402

# Round 6: 1.25 x the worst error measured over weight seeds 0 (the HF goldens), 1 and 2 (against the oracle, pinned to HF's
# CLIPTextModel at 3e-6 by oracle/make_golden_clip.py).  Measured (pass I, MI355X): 3.13e-3 (sd_b2, seed 0), 3.67e-3 (seed 1), 3.42e-3 (seed 2),
# rms 6.4 - 7.6e-4 on a LayerNorm output of rms 1 -> 1.25 x 3.67e-3.  (Rounds 1-5 carried 4e-3 on seed 0 alone.)
CLIP_TOL = 4.6e-3
CFGS = {'tiny': clip_ref.TINY_CLIP, 'sd': clip_ref.SD_CLIP}
_models = {}


def _text_config(cfg):
    return dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, max_position_embeddings=cfg.max_positions)


class _FakeTokenizer:
    """bytes of the prompt -> ids, BOS / EOS-padding like CLIP's tokenizer (no vocabulary files offline)"""

    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, text, truncation=True, max_length=77, padding='max_length', return_tensors='pt', **kw):
        text = [text] if isinstance(text, str) else list(text)
        ids = torch.full((len(text), max_length), self.vocab - 1, dtype=torch.long)
        for i, s in enumerate(text):
            toks = [self.vocab - 2] + [(b * 37) % (self.vocab - 2) for b in s.encode()][:max_length - 2] + [self.vocab - 1]
            ids[i, :len(toks)] = torch.tensor(toks)
        return {'input_ids': ids}


def _model(cfg_name, wseed):
    key = (cfg_name, wseed)
    if key not in _models:
        _models.clear()
        torch.cuda.empty_cache()
        from stable_diffusion_amd import FrozenCLIPEmbedderHIP
        cfg = CFGS[cfg_name]
        sd = clip_ref.make_clip_state_dict(cfg, wseed)
        m = FrozenCLIPEmbedderHIP(text_config=_text_config(cfg), tokenizer=_FakeTokenizer(cfg.vocab_size))
        missing, unexpected = m.load_state_dict({'transformer.' + k: v for k, v in sd.items()}, strict=False)
        assert not unexpected and missing == ['transformer.text_model.embeddings.position_ids']
        _models[key] = (m.cuda(), sd)
    return _models[key]


@pytest.mark.parametrize('case', ['tiny_b2', 'tiny_b3_L40', 'sd_b2'])
def test_clip_matches_hf_golden(case, golden_dir):
    z = np.load(os.path.join(golden_dir, f'clip_{case}.npz'))
    cfg_name = str(z['cfg'])
    cfg = CFGS[cfg_name]
    m, sd = _model(cfg_name, int(z['weight_seed']))
    ids = clip_ref.make_clip_ids(cfg, int(z['batch']), int(z['L']), seed=int(z['input_seed']))
    out = m.encode_ids(ids.cuda())
    torch.cuda.synchronize()
    ref = torch.from_numpy(z['out'])
    err = (out.float().cpu() - ref).abs()
    print(f'[clip {case}] HIP-vs-HF(fp32) max-abs {err.max():.3e} rms {err.pow(2).mean().sqrt():.3e} |x|max {ref.abs().max():.3f} '
          f'nan={bool(torch.isnan(out).any())}', flush=True)
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert float(err.max()) <= CLIP_TOL


@pytest.mark.parametrize('wseed', [1, 2])
def test_clip_other_weight_seeds(wseed):
    """The SD text model under weight seeds the goldens do not use, against the oracle restatement on the same token ids."""
    cfg = clip_ref.SD_CLIP
    m, sd = _model('sd', wseed)
    ids = clip_ref.make_clip_ids(cfg, 2, 77, seed=5)
    ref = clip_ref.clip_text_forward(sd, cfg, ids)
    err = (m.encode_ids(ids.cuda()).float().cpu() - ref).abs()
    print(f'[clip sd_b2 weight seed {wseed}] HIP-vs-oracle(fp32) max-abs {err.max():.3e} rms {err.pow(2).mean().sqrt():.3e}', flush=True)
    assert float(err.max()) <= CLIP_TOL


def test_clip_is_causal_repeatable_and_batch_independent():
    cfg = clip_ref.TINY_CLIP
    m, sd = _model('tiny', 0)
    ids = clip_ref.make_clip_ids(cfg, 4, 77, seed=3).cuda()
    a = m.encode_ids(ids)
    assert torch.equal(a, m.encode_ids(ids))
    # causal: changing token j leaves every position < j untouched (bit for bit: same rows, same arithmetic)
    ids2 = ids.clone(); ids2[:, 50] = (ids2[:, 50] + 1) % (cfg.vocab_size - 2)
    b = m.encode_ids(ids2)
    assert torch.equal(a[:, :50], b[:, :50]) and not torch.equal(a[:, 50:], b[:, 50:])
    one = m.encode_ids(ids[1:2])
    assert (one - a[1:2]).abs().max().item() <= CLIP_TOL
    ref = clip_ref.clip_text_forward(sd, cfg, ids.cpu())
    assert (a.cpu() - ref).abs().max().item() <= CLIP_TOL


def test_clip_text_interface_and_refusals():
    cfg = clip_ref.TINY_CLIP
    m, sd = _model('tiny', 0)
    z = m.encode(['a photograph of an astronaut riding a horse', ''])        # FrozenCLIPEmbedder.encode(text), modules.py:161
    assert z.shape == (2, 77, cfg.hidden_size) and torch.isfinite(z).all()
    assert torch.equal(z, m(['a photograph of an astronaut riding a horse', '']))
    with pytest.raises(RuntimeError, match='no CPU'):
        m.encode_ids(torch.zeros(1, 77, dtype=torch.long))
    with pytest.raises(IndexError):
        m.encode_ids(torch.full((1, 77), cfg.vocab_size, dtype=torch.long, device='cuda'))
    with pytest.raises(ValueError):
        m.encode_ids(torch.zeros(1, 78, dtype=torch.long, device='cuda'))
