"""Compile the reference's Python sources -- where they lie under /root/reference -- into a bytecode bundle under
`oracle/_ref/refbundle/`, so that the UNMODIFIED `scripts/txt2img.py` / `scripts/img2img.py` can drive the MI355X path on a
GPU box, where /root/reference does not exist.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

    python oracle/build_ref_bundle.py [--reference /root/reference] [--out oracle/_ref/refbundle]

The rule this follows is the one for a compiled reference: a recipe committed under `oracle/`, sources read where they lie,
outputs only into `oracle/_ref/` (git-ignored -- nothing of the reference enters the history -- but not gpurun-ignored, so
the bundle travels to the GPU box like the built `.so` files).  What is written:

  * `ldm/**/*.pyc`, `scripts/txt2img.pyc`, `scripts/img2img.pyc` -- `py_compile` output in the "sourceless" layout
    (`module.pyc` where `module.py` would be), importable with the bundle root on `sys.path`.  No source text is copied;
  * `v1-inference.json` -- the PARSED content of `configs/stable-diffusion/v1-inference.yaml` (a dict of hyper-parameters);
    `tools/run_reference_script.py` writes its patched yaml from it;
  * `MANIFEST.json` -- per file: reference path, sha256 of the source it was compiled from, python version.

The product never reads the bundle: only `tools/run_reference_script.py` (a launcher for the reference's own scripts) and the
GPU test that calls it (`tests/test_reference_script_gpu.py`) do.
"""
import argparse
import hashlib
import json
import os
import py_compile
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_OUT = os.path.join(ROOT, 'oracle', '_ref', 'refbundle')
SCRIPTS = ('txt2img.py', 'img2img.py')


def build(reference='/root/reference', out=DEFAULT_OUT, verbose=True):
    reference = os.path.abspath(reference)
    if not os.path.isdir(os.path.join(reference, 'ldm')):
        raise SystemExit(f'reference checkout not found at {reference}')
    import yaml
    if os.path.isdir(out):
        shutil.rmtree(out)
    os.makedirs(out)
    manifest = {'python': sys.version.split()[0], 'reference': reference, 'files': {}}
    jobs = []
    for dirpath, dirnames, filenames in os.walk(os.path.join(reference, 'ldm')):
        dirnames[:] = [d for d in dirnames if d != '__pycache__']
        for fn in sorted(filenames):
            if fn.endswith('.py'):
                src = os.path.join(dirpath, fn)
                jobs.append((src, os.path.relpath(src, reference)))
    for fn in SCRIPTS:
        jobs.append((os.path.join(reference, 'scripts', fn), os.path.join('scripts', fn)))
    for src, rel in jobs:
        dst = os.path.join(out, rel[:-3] + '.pyc')
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile = the path tracebacks show: the reference's own (so a failure on the GPU box names the reference line)
        py_compile.compile(src, cfile=dst, dfile=os.path.join('/root/reference', rel), doraise=True)
        with open(src, 'rb') as f:
            manifest['files'][rel] = hashlib.sha256(f.read()).hexdigest()
    with open(os.path.join(reference, 'configs', 'stable-diffusion', 'v1-inference.yaml')) as f:
        cfg = yaml.safe_load(f)
    with open(os.path.join(out, 'v1-inference.json'), 'w') as f:
        json.dump(cfg, f, indent=1, sort_keys=True)
    with open(os.path.join(out, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    if verbose:
        print(f'reference bytecode bundle: {len(jobs)} modules -> {out}')
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--reference', default=os.environ.get('SD_REFERENCE', '/root/reference'))
    ap.add_argument('--out', default=DEFAULT_OUT)
    a = ap.parse_args()
    build(a.reference, a.out)
