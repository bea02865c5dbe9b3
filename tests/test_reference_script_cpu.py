"""BASELINE.json configs[0] as an automated check: the reference's UNMODIFIED scripts/txt2img.py, CPU, fp32 -- the reference plumbing
run (`scripts/txt2img.py SD-v1-4 256x256, 10 DDIM steps, batch 1, PyTorch CPU float32`), shrunk to 64 x 64 / 2 steps so that it fits
the CPU suite.  Nothing of this repository's compute path runs here (no GPU, no libsdmi): what it pins is that
`tools/run_reference_script.py` can execute the script's own code object -- argument parser, `instantiate_from_config` +
`load_state_dict` of the full SD-v1 module tree (synthetic weights under the checkpoint's key names), the DDIM loop,
`decode_first_stage`, clamp, safety check, PNG writer -- which is the harness the GPU test (tests/test_reference_script_gpu.py)
then points at the HIP classes.  /root/reference (build container) or the bytecode bundle oracle/_ref/refbundle must be present."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNDLE = os.path.join(ROOT, 'oracle', '_ref', 'refbundle')
REFERENCE = os.environ.get('SD_REFERENCE', '/root/reference')


def _bundle_usable():
    """the bundle is .pyc files: bound to the interpreter minor version that compiled them (MANIFEST.json records it)"""
    man = os.path.join(BUNDLE, 'MANIFEST.json')
    if not os.path.isdir(os.path.join(BUNDLE, 'ldm')) or not os.path.exists(man):
        return False
    try:
        py = str(json.load(open(man)).get('python', ''))
    except Exception:
        return False
    return py.split('.')[:2] == [str(sys.version_info[0]), str(sys.version_info[1])]


HAVE_REF = os.path.isdir(os.path.join(REFERENCE, 'ldm')) or _bundle_usable()


@pytest.mark.skipif(not HAVE_REF, reason='neither a reference checkout nor a bytecode bundle for this interpreter')
@pytest.mark.timeout(900)
def test_configs0_unmodified_txt2img_on_cpu_fp32(tmp_path):
    from PIL import Image
    out = tmp_path / 'cfg0'
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'run_reference_script.py'), '--reference', REFERENCE, '--offline-stubs', 'txt2img',
           '--', '--ckpt', 'synthetic', '--n_samples', '1', '--n_iter', '1', '--ddim_steps', '2', '--H', '64', '--W', '64',
           '--precision', 'full', '--seed', '7', '--prompt', 'a photograph of an astronaut riding a horse', '--outdir', str(out)]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')     # the CPU run, also on a GPU box
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env)
    assert r.returncode == 0, f'txt2img.py (CPU) failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}'
    img = np.asarray(Image.open(out / 'samples' / '00000.png'))
    assert img.shape == (64, 64, 3) and img.dtype == np.uint8 and img.std() > 1.0
    assert os.path.exists(out / 'grid-0000.png')
