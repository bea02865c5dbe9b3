"""Latency of the text encoder for the (uc, c) prompt pair: FrozenCLIPEmbedderHIP vs Hugging Face CLIPTextModel on
PyTorch-ROCm (fp32 and fp16 autocast), seeded random weights in the ViT-L/14 text-tower architecture.  python tools/prof_clip.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import transformers  # noqa: E402

from stable_diffusion_amd import FrozenCLIPEmbedderHIP  # noqa: E402
from stable_diffusion_amd.clip import CLIP_VIT_L14_TEXT  # noqa: E402
from stable_diffusion_amd.synthetic import synthetic_clip_state_dict  # noqa: E402

dev = torch.device('cuda')
sd = synthetic_clip_state_dict(None, 0)
m = FrozenCLIPEmbedderHIP(tokenizer=object())
m.load_state_dict(sd, strict=True)
m = m.to(dev)
ids = torch.randint(0, 49000, (2, 77), device=dev)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_hip = timeit(lambda: m.encode_ids(ids))
hf = transformers.CLIPTextModel(transformers.CLIPTextConfig(**CLIP_VIT_L14_TEXT, hidden_act='quick_gelu')).eval().to(dev)
own = hf.state_dict()
prefixed = any(k.startswith('text_model.') for k in own)
mapped = {}
for k, v in sd.items():
    k = k[len('transformer.'):]
    k = k if prefixed else k[len('text_model.'):]
    if k in own:
        mapped[k] = v
hf.load_state_dict(mapped, strict=False)
with torch.no_grad():
    ref = hf(input_ids=ids).last_hidden_state
    out = m.encode_ids(ids)
    t_f32 = timeit(lambda: hf(input_ids=ids).last_hidden_state)
    with torch.autocast('cuda', dtype=torch.float16):
        t_f16 = timeit(lambda: hf(input_ids=ids).last_hidden_state)
        ac = hf(input_ids=ids).last_hidden_state
print(f'text encoder, 2 x 77 tokens: libsdmi {t_hip:.3f} ms | transformers {transformers.__version__} on PyTorch-ROCm fp32 {t_f32:.3f} ms, '
      f'fp16 autocast {t_f16:.3f} ms')
print(f'max-abs vs PyTorch-ROCm fp32: libsdmi {(out - ref).abs().max().item():.3e} | fp16 autocast {(ac.float() - ref).abs().max().item():.3e}')
