"""The reference's UNMODIFIED scripts/txt2img.py and scripts/img2img.py driving the MI355X path (north_star:
"so scripts/txt2img.py still drives it unchanged").

The GPU boxes have no /root/reference; what travels with the repo snapshot is the bytecode bundle that
`oracle/build_ref_bundle.py` compiled from the reference where it lies (oracle/_ref/refbundle/, git-ignored; built by
`__graft_entry__.build()` whenever /root/reference is visible).  `tools/run_reference_script.py --hip` executes the script's
own code object with three `target:` strings of the inference yaml pointing at this package and the HIP samplers in place of
`PLMSSampler` / `DDIMSampler` -- nothing else of the script changes: its argument parser, its model loader
(`instantiate_from_config` + `load_state_dict`), its sampling loop, `decode_first_stage`, clamp, safety check, PNG writer.

There is no SD checkpoint, CLIP tokenizer or safety-checker weight file in the environment (no network): `--ckpt synthetic`
and `--offline-stubs` provide seeded stand-ins for exactly those (each announced on stderr), so the images are noise-like --
the test checks the plumbing and the numbers, not the picture:
  * the script finishes and writes the sample + grid PNGs at the requested size;
  * every UNet / first-stage / text-encoder forward went through libsdmi, as often as the script's arguments imply (call counts
    reported by the launcher: 10 PLMS steps = 11 UNet calls, strength 0.5 of 10 DDIM steps = 5).
The numbers of that sequence of calls are checked elsewhere (tests/test_pipeline_gpu.py against the oracle pipeline).
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNDLE = os.path.join(ROOT, 'oracle', '_ref', 'refbundle')
pytestmark = pytest.mark.gpu


def _bundle_usable():
    """.pyc files are bound to the interpreter minor version that compiled them (MANIFEST.json records it): a different one would die
    with 'bad magic number' instead of skipping"""
    import json
    man = os.path.join(BUNDLE, 'MANIFEST.json')
    if not os.path.isdir(os.path.join(BUNDLE, 'ldm')) or not os.path.exists(man):
        return False
    try:
        py = str(json.load(open(man)).get('python', ''))
    except Exception:
        return False
    return py.split('.')[:2] == [str(sys.version_info[0]), str(sys.version_info[1])]


NO_BUNDLE = 'no reference bytecode bundle for this interpreter (oracle/build_ref_bundle.py needs /root/reference at build time)'


def _run(script, extra, outdir):
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'run_reference_script.py'), '--reference', '/nonexistent-on-purpose',
           '--hip', '--offline-stubs', script, '--', '--ckpt', 'synthetic', '--n_samples', '1', '--n_iter', '1',
           '--ddim_steps', '10', '--seed', '7', '--prompt', 'a photograph of an astronaut riding a horse',
           '--outdir', str(outdir)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f'{script}.py failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}'
    return r.stdout + r.stderr


def _calls(log):
    m = re.search(r'libsdmi calls -- UNetModelHIP.forward x(\d+), AutoencoderKLHIP.decode x(\d+), FrozenCLIPEmbedderHIP.forward x(\d+)', log)
    assert m, log[-2000:]
    return tuple(int(g) for g in m.groups())


@pytest.mark.skipif(not _bundle_usable(), reason=NO_BUNDLE)
def test_unmodified_txt2img_script_drives_the_hip_path(tmp_path):
    from PIL import Image
    out = tmp_path / 'txt2img'
    log = _run('txt2img', ['--H', '256', '--W', '256', '--plms'], out)
    unet_calls, vae_calls, clip_calls = _calls(log)
    assert unet_calls == 11, log[-1500:]        # 10 PLMS steps = 11 UNet calls (the first step evaluates twice), CFG pair batched
    assert vae_calls == 1 and clip_calls == 2   # uc and c
    img = np.asarray(Image.open(out / 'samples' / '00000.png'))
    assert img.shape == (256, 256, 3) and img.dtype == np.uint8
    assert img.std() > 1.0                       # not a constant frame
    assert os.path.exists(out / 'grid-0000.png')


@pytest.mark.skipif(not _bundle_usable(), reason=NO_BUNDLE)
def test_unmodified_img2img_script_drives_the_hip_path(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    init = tmp_path / 'init.png'
    Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(init)
    out = tmp_path / 'img2img'
    log = _run('img2img', ['--init-img', str(init), '--strength', '0.5'], out)
    unet_calls, vae_calls, clip_calls = _calls(log)
    assert unet_calls == 5, log[-1500:]         # strength 0.5 of 10 DDIM steps
    assert vae_calls == 1 and clip_calls == 2
    img = np.asarray(Image.open(out / 'samples' / '00000.png'))
    assert img.shape == (256, 256, 3) and img.std() > 1.0
