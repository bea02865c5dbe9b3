"""Generate tests/golden/clip_*.npz from Hugging Face `CLIPTextModel` (build container only).

    PYTHONPATH=/root/repo python oracle/make_golden_clip.py

`FrozenCLIPEmbedder` (ldm/modules/encoders/modules.py:137-162) is a thin wrapper around transformers' CLIPTextModel; the
reference pins transformers==4.19.2, the build container has a newer release whose CLIPTextModel computes the same
function but names its parameters without the `text_model.` prefix.  This script builds that model from a
`CLIPTextConfig` (no download), loads `oracle.clip_ref.make_clip_state_dict` into it (prefix mapped, strict=True),
asserts the oracle restatement equals it and stores its `last_hidden_state` as fixtures.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import transformers
    from oracle import clip_ref
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    cases = [('tiny_b2', clip_ref.TINY_CLIP, 0, 2, 77), ('tiny_b3_L40', clip_ref.TINY_CLIP, 1, 3, 40),
             ('sd_b2', clip_ref.SD_CLIP, 0, 2, 77)]
    for name, cfg, seed, b, L in cases:
        hf_cfg = transformers.CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                             intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_layers,
                                             num_attention_heads=cfg.num_heads, max_position_embeddings=cfg.max_positions,
                                             hidden_act='quick_gelu', attention_dropout=0.0)
        m = transformers.CLIPTextModel(hf_cfg).eval()
        sd = clip_ref.make_clip_state_dict(cfg, seed)
        own = set(m.state_dict().keys())
        prefixed = any(k.startswith('text_model.') for k in own)
        mapped = {(k if prefixed else k[len('text_model.'):]): v for k, v in sd.items()}
        extra = {k: v for k, v in m.state_dict().items() if k not in mapped}          # e.g. a position_ids buffer
        assert all('position_ids' in k for k in extra), extra.keys()
        m.load_state_dict({**mapped, **extra}, strict=True)
        ids = clip_ref.make_clip_ids(cfg, b, L, seed=1)
        with torch.no_grad():
            ref = m(input_ids=ids).last_hidden_state
        orc = clip_ref.clip_text_forward(sd, cfg, ids)
        err = (ref - orc).abs().max().item()
        print(f'[clip {name}] transformers {transformers.__version__}: out {tuple(ref.shape)} |x| max {ref.abs().max():.3f} '
              f'rms {ref.pow(2).mean().sqrt():.3f} oracle-vs-HF {err:.3e}', flush=True)
        assert err < 5e-5, err
        np.savez_compressed(os.path.join(out_dir, f'clip_{name}.npz'), out=ref.numpy().astype(np.float32),
                            cfg='tiny' if cfg is clip_ref.TINY_CLIP else 'sd', weight_seed=seed, input_seed=1, batch=b, L=L,
                            transformers_version=transformers.__version__, oracle_vs_hf=err)
    print('CLIP golden fixtures written to', out_dir)


if __name__ == '__main__':
    main()
