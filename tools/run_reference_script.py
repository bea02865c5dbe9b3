"""Run the reference's UNMODIFIED `scripts/txt2img.py` / `scripts/img2img.py` offline, optionally on the MI355X path.

    python tools/run_reference_script.py [--hip] [--reference /root/reference] txt2img -- --plms --ddim_steps 50 ...

What this launcher does (nothing in the reference tree is edited or copied):
  * puts minimal stand-ins into `sys.modules` for the third-party imports the scripts need but THIS box lacks
    (SURVEY.md 8b): omegaconf, pytorch_lightning, torchvision.utils, cv2, imwatermark, diffusers' safety checker,
    taming, clip, kornia -- each only when the real package fails to import;
  * the HF `from_pretrained` calls (CLIP tokenizer / text model, safety feature extractor) and the safety checker try
    the REAL thing first; only when that fails (no network / no cache) AND `--offline-stubs` was given do they fall back
    to seeded random-init stand-ins, each with a loud warning.  With a real `--ckpt` path the stand-ins are refused:
    a byte-hash tokenizer and a disabled safety checker must never run silently beside real weights;
  * `--ckpt synthetic[:seed]` makes `torch.load` return a seeded random UNet + first-stage state_dict under the
    checkpoint's key names (`model.diffusion_model.*`, `first_stage_model.*`) -- there is no SD checkpoint in the
    environment; a real `--ckpt path` is loaded as usual;
  * `--hip`: writes a patched copy of the reference's `v1-inference.yaml` (only `unet_config.target` changed to
    `stable_diffusion_amd.unet.UNetModelHIP`, `first_stage_config.target` to
    `stable_diffusion_amd.vae.AutoencoderKLHIP` and `cond_stage_config.target` to
    `stable_diffusion_amd.clip.FrozenCLIPEmbedderHIP`) to a temp file, passes it as `--config`, and swaps
    `ldm.models.diffusion.plms.PLMSSampler` / `ddim.DDIMSampler` / `dpm_solver.DPMSolverSampler` for the HIP samplers
    before the script imports them;
  * on a GPU-less host (BASELINE.json configs[0], the CPU plumbing check) it neutralises the hard-coded
    `.cuda()` / `torch.device("cuda")` uses (`txt2img.py:64`, `plms.py:18-22`); use `--precision full` there.
Then it `runpy`-executes the script with the remaining arguments.
"""
import argparse
import os
import runpy
import sys
import tempfile
import types

import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _module(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition('.')
    if parent:
        if parent not in sys.modules:
            _module(parent)
        setattr(sys.modules[parent], leaf, m)
    return m


class AttrDict(dict):
    """dict with attribute access (what the scripts use of an OmegaConf node)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__


def _wrap(x):
    if isinstance(x, dict):
        return AttrDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return ListConfig(_wrap(v) for v in x)
    return x


class ListConfig(list):
    pass


def _warn(msg):
    print('\n' + '!' * 100 + f'\n!! run_reference_script: {msg}\n' + '!' * 100 + '\n', file=sys.stderr, flush=True)


def install_stubs(have_gpu, offline_stubs=False, real_ckpt=False):
    import yaml
    import transformers          # before the stand-ins: it probes optional packages (torchvision, ...) via find_spec

    class OmegaConf:
        @staticmethod
        def load(path):
            with open(path) as f:
                return _wrap(yaml.safe_load(f))

        @staticmethod
        def to_container(x, resolve=True):
            return x
    if 'omegaconf' not in sys.modules:
        try:
            import omegaconf  # noqa: F401
        except ImportError:
            _module('omegaconf', OmegaConf=OmegaConf)
            _module('omegaconf.listconfig', ListConfig=ListConfig)

    try:
        import pytorch_lightning  # noqa: F401
    except ImportError:
        class LightningModule(nn.Module):
            @property
            def device(self):
                try:
                    return next(self.parameters()).device
                except StopIteration:
                    return torch.device('cpu')

            def log(self, *a, **k): pass
            def log_dict(self, *a, **k): pass

        def seed_everything(seed=None):
            import random
            import numpy as np
            random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
            return seed
        _module('pytorch_lightning', LightningModule=LightningModule, seed_everything=seed_everything,
                Callback=object, Trainer=object)
        _module('pytorch_lightning.utilities')
        _module('pytorch_lightning.utilities.distributed', rank_zero_only=lambda f: f)

    try:
        import torchvision  # noqa: F401
    except ImportError:
        def make_grid(t, nrow=8, padding=2, **kw):
            n, c, h, w = t.shape
            ncol = min(nrow, n)
            nr = (n + ncol - 1) // ncol
            grid = t.new_zeros((c, nr * (h + padding) + padding, ncol * (w + padding) + padding))
            for i in range(n):
                r, cc = divmod(i, ncol)
                y, x = padding + r * (h + padding), padding + cc * (w + padding)
                grid[:, y:y + h, x:x + w] = t[i]
            return grid
        _module('torchvision')
        _module('torchvision.utils', make_grid=make_grid)

    try:
        import cv2  # noqa: F401
    except ImportError:
        _module('cv2', COLOR_RGB2BGR=4, cvtColor=lambda img, code: img[:, :, ::-1].copy())
    try:
        import imwatermark  # noqa: F401
    except ImportError:
        class WatermarkEncoder:
            def set_watermark(self, *a, **k): pass
            def encode(self, img, method): return img
        _module('imwatermark', WatermarkEncoder=WatermarkEncoder)

    def _stand_in_allowed(what):
        """The stand-ins for network-bound pieces are opt-in and never combined with a real checkpoint."""
        if real_ckpt:
            raise SystemExit(f'{what} is not available offline and a real --ckpt was given: refusing to substitute a '
                             'stand-in (prompts would be tokenized into garbage / the safety filter would be off). '
                             'Provide the HF cache, or use --ckpt synthetic with --offline-stubs.')
        if not offline_stubs:
            raise SystemExit(f'{what} is not available offline; pass --offline-stubs to run with a seeded stand-in '
                             '(synthetic checkpoints only)')

    class _Safety:
        @classmethod
        def from_pretrained(cls, *a, **k): return cls()
        def __call__(self, images, clip_input): return images, [False] * len(images)
    try:
        import diffusers.pipelines.stable_diffusion.safety_checker as _sc   # noqa: F401
        _real_sc = _sc.StableDiffusionSafetyChecker.from_pretrained.__func__

        def _safety_from_pretrained(cls, *a, **k):
            try:
                return _real_sc(cls, *a, **k)
            except Exception as e:      # no network / no cache
                _stand_in_allowed(f'the NSFW safety checker weights ({type(e).__name__})')
                _warn('SAFETY CHECKER STUBBED: every image is reported as safe (has_nsfw = False)')
                return _Safety()
        _sc.StableDiffusionSafetyChecker.from_pretrained = classmethod(_safety_from_pretrained)
    except ImportError:
        _stand_in_allowed('the `diffusers` package (NSFW safety checker)')
        _warn('SAFETY CHECKER STUBBED (diffusers is not installed): every image is reported as safe')
        _module('diffusers.pipelines.stable_diffusion.safety_checker', StableDiffusionSafetyChecker=_Safety)

    for name in ('clip', 'kornia', 'taming', 'taming.modules', 'taming.modules.vqvae'):
        if name not in sys.modules:
            _module(name)
    _module('taming.modules.vqvae.quantize', VectorQuantizer2=type('VectorQuantizer2', (nn.Module,), {}))

    # ---- HF from_pretrained: the real call first; a seeded stand-in only offline, opt-in, with a warning ----------
    import transformers

    def _try_real(cls_name, what, make_stub):
        real = getattr(transformers, cls_name).from_pretrained.__func__

        def from_pretrained(cls, *a, **k):
            try:
                return real(cls, *a, **k)
            except Exception as e:      # no network / nothing in the HF cache
                _stand_in_allowed(f'{what} ({type(e).__name__}: HF hub unreachable and not cached)')
                _warn(f'{what.upper()} STUBBED with a seeded stand-in -- outputs are NOT those of the real model')
                return make_stub()
        getattr(transformers, cls_name).from_pretrained = classmethod(from_pretrained)

    class _FeatureExtractor:
        def __call__(self, images, return_tensors='pt'):
            return types.SimpleNamespace(pixel_values=torch.zeros(len(images), 3, 224, 224))

    class _Tokenizer:
        """deterministic stand-in: bytes of the prompt -> ids, BOS/EOS/pad like CLIP's (49406 / 49407)."""
        def __call__(self, text, truncation=True, max_length=77, padding='max_length', return_tensors='pt', **kw):
            if isinstance(text, str):
                text = [text]
            ids = torch.full((len(text), max_length), 49407, dtype=torch.long)
            for i, s in enumerate(text):
                toks = [49406] + [1000 + (b * 37) % 40000 for b in s.encode()][:max_length - 2] + [49407]
                ids[i, :len(toks)] = torch.tensor(toks)
            return {'input_ids': ids}

    def _clip_text():
        cfg = transformers.CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                          num_attention_heads=12, max_position_embeddings=77, hidden_act='quick_gelu')
        torch.manual_seed(1234)
        return transformers.CLIPTextModel(cfg)
    _try_real('AutoFeatureExtractor', 'the safety feature extractor', _FeatureExtractor)
    _try_real('CLIPTokenizer', 'the CLIP tokenizer', _Tokenizer)
    _try_real('CLIPTextModel', 'the CLIP text model weights', _clip_text)

    if not have_gpu:   # configs[0]: CPU plumbing run
        nn.Module.cuda = lambda self, device=None: self


HIP_COND_STAGE = False
BUNDLE = os.path.join(REPO, 'oracle', '_ref', 'refbundle')     # oracle/build_ref_bundle.py (git-ignored build output)


def _inference_yaml_text(ref):
    """configs/stable-diffusion/v1-inference.yaml of a checkout, or the same content re-serialised from the bundle's parsed copy."""
    src = os.path.join(ref, 'configs', 'stable-diffusion', 'v1-inference.yaml')
    if os.path.exists(src):
        return open(src).read()
    import json
    import yaml
    with open(os.path.join(ref, 'v1-inference.json')) as f:
        return yaml.safe_dump(json.load(f), default_flow_style=False, sort_keys=False)


def _report_hip_calls():
    """--hip: count the forwards that reach libsdmi and say so at exit (the evidence that the script drove the HIP path)."""
    import atexit
    from stable_diffusion_amd import unet as _u, vae as _v, clip as _c
    counts = {'UNetModelHIP.forward': 0, 'AutoencoderKLHIP.decode': 0, 'FrozenCLIPEmbedderHIP.forward': 0}

    def wrap(cls, name, key):
        real = getattr(cls, name)

        def counted(self, *a, **k):
            counts[key] += 1
            return real(self, *a, **k)
        setattr(cls, name, counted)
    wrap(_u.UNetModelHIP, 'forward', 'UNetModelHIP.forward')
    wrap(_v.AutoencoderKLHIP, 'decode', 'AutoencoderKLHIP.decode')
    wrap(_c.FrozenCLIPEmbedderHIP, 'forward', 'FrozenCLIPEmbedderHIP.forward')

    def report():
        from stable_diffusion_amd import _lib
        print('run_reference_script: libsdmi calls -- ' + ', '.join(f'{k} x{v}' for k, v in counts.items()) +
              f' (library: {_lib.LIB_PATH})', flush=True)
    atexit.register(report)


def patch_torch_load():
    real = torch.load

    def load(f, *a, **k):
        if isinstance(f, str) and f.startswith('synthetic'):
            seed = int(f.split(':')[1]) if ':' in f else 0
            from stable_diffusion_amd.synthetic import (SD_V1_UNET_KWARGS, SD_V1_VAE_DDCONFIG, synthetic_clip_state_dict,
                                                        synthetic_state_dict, synthetic_vae_state_dict)
            sd = {'model.diffusion_model.' + key: v for key, v in synthetic_state_dict(SD_V1_UNET_KWARGS, seed).items()}
            sd.update({'first_stage_model.' + key: v
                       for key, v in synthetic_vae_state_dict(SD_V1_VAE_DDCONFIG, 4, seed).items()})
            if HIP_COND_STAGE:     # the reference's own FrozenCLIPEmbedder gets its (stand-in) weights from from_pretrained
                sd.update({'cond_stage_model.' + key: v for key, v in synthetic_clip_state_dict(None, seed).items()})
            return {'state_dict': sd}
        return real(f, *a, **k)
    torch.load = load


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--reference', default=os.environ.get('SD_REFERENCE', '/root/reference'))
    ap.add_argument('--hip', action='store_true', help='UNetModelHIP + HIP samplers (needs the MI355X)')
    ap.add_argument('--offline-stubs', action='store_true',
                    help='allow seeded stand-ins for the CLIP tokenizer / text model / safety checker when the HF hub is '
                         'unreachable (synthetic checkpoints only; each substitution prints a warning)')
    ap.add_argument('script', choices=['txt2img', 'img2img'])
    ap.add_argument('rest', nargs=argparse.REMAINDER)
    args = ap.parse_args()
    rest = args.rest[1:] if args.rest[:1] == ['--'] else args.rest
    ref = os.path.abspath(args.reference)
    if not os.path.isdir(os.path.join(ref, 'ldm')) and os.path.isdir(os.path.join(BUNDLE, 'ldm')):
        # a GPU box: no reference checkout, but the bytecode bundle oracle/build_ref_bundle.py compiled from it travelled with the repo
        print(f'run_reference_script: no reference checkout at {ref}; using the bytecode bundle {BUNDLE}', file=sys.stderr)
        ref = BUNDLE
        try:
            import json
            built_for = str(json.load(open(os.path.join(BUNDLE, 'MANIFEST.json'))).get('python', ''))
        except Exception:
            built_for = ''
        here = f'{sys.version_info[0]}.{sys.version_info[1]}'
        if built_for and built_for.split('.')[:2] != here.split('.'):
            raise SystemExit(f'the bytecode bundle was compiled by Python {built_for}, this is Python {here}: rebuild it '
                             '(python oracle/build_ref_bundle.py) with this interpreter')
    assert os.path.isdir(os.path.join(ref, 'ldm')), f'reference checkout not found at {ref} (and no bundle at {BUNDLE})'
    sys.path.insert(0, ref)
    have_gpu = torch.cuda.is_available()
    if args.hip and not have_gpu:
        raise SystemExit('--hip needs the MI355X (the HIP path has no CPU fallback)')
    ckpt = rest[rest.index('--ckpt') + 1] if '--ckpt' in rest and rest.index('--ckpt') + 1 < len(rest) else None
    real_ckpt = not (ckpt or '').startswith('synthetic')     # the script's default --ckpt is a real file as well
    install_stubs(have_gpu, offline_stubs=args.offline_stubs, real_ckpt=real_ckpt)
    patch_torch_load()

    import ldm.models.diffusion.ddim as ddim
    import ldm.models.diffusion.plms as plms
    if args.hip:
        from stable_diffusion_amd import DDIMSamplerHIP, DPMSolverSamplerHIP, PLMSSamplerHIP
        import ldm.models.diffusion.dpm_solver as dpm
        plms.PLMSSampler, ddim.DDIMSampler, dpm.DPMSolverSampler = PLMSSamplerHIP, DDIMSamplerHIP, DPMSolverSamplerHIP
        text = _inference_yaml_text(ref)
        old = 'target: ldm.modules.diffusionmodules.openaimodel.UNetModel'
        assert old in text
        tmp = tempfile.NamedTemporaryFile('w', suffix='-mi355x.yaml', delete=False)
        global HIP_COND_STAGE
        HIP_COND_STAGE = True
        old_clip = 'target: ldm.modules.encoders.modules.FrozenCLIPEmbedder'
        assert old_clip in text
        text = text.replace(old_clip, 'target: stable_diffusion_amd.clip.FrozenCLIPEmbedderHIP')
        old_vae = 'target: ldm.models.autoencoder.AutoencoderKL'
        assert old_vae in text
        text = text.replace(old_vae, 'target: stable_diffusion_amd.vae.AutoencoderKLHIP')
        tmp.write(text.replace(old, 'target: stable_diffusion_amd.unet.UNetModelHIP'))
        tmp.close()
        rest = ['--config', tmp.name] + rest
    elif not have_gpu:
        import ldm.models.diffusion.dpm_solver.sampler as dpm_s
        for cls in (plms.PLMSSampler, ddim.DDIMSampler, dpm_s.DPMSolverSampler):   # plms.py:18-22 hard-codes torch.device("cuda")
            cls.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    if not have_gpu:                                            # encoders/modules.py:139 defaults device="cuda"
        import ldm.modules.encoders.modules as enc
        _init = enc.FrozenCLIPEmbedder.__init__

        def _cpu_init(self, *a, **k):
            k.setdefault('device', 'cpu')
            _init(self, *a, **k)
        enc.FrozenCLIPEmbedder.__init__ = _cpu_init
    if '--config' not in rest:
        plain = os.path.join(ref, 'configs', 'stable-diffusion', 'v1-inference.yaml')
        if not os.path.exists(plain):       # the bundle holds the parsed config only
            tmp = tempfile.NamedTemporaryFile('w', suffix='-v1-inference.yaml', delete=False)
            tmp.write(_inference_yaml_text(ref))
            tmp.close()
            plain = tmp.name
        rest = ['--config', plain] + rest
    script = os.path.join(ref, 'scripts', args.script + '.py')
    if not os.path.exists(script):
        script += 'c'                       # bytecode bundle: runpy executes a compiled file the same way
    sys.argv = [script] + rest
    if args.hip:
        _report_hip_calls()
    os.chdir(ref if (os.access(ref, os.W_OK) and ref != BUNDLE) else tempfile.mkdtemp())
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
