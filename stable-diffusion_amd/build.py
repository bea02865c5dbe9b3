"""Build libsdmi.so (hipcc, gfx950) in-tree.  `python stable-diffusion_amd/build.py [--force]`.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
`stable-diffusion_amd/libsdmi.so` travels to the GPU box with the repo snapshot.
"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# A second library next to the product one (instrumented builds, two-build A/Bs through SDMI_LIB_PATH):
#   SDMI_CXXFLAGS='-DSDMI_IGEMM_TIMING' SDMI_LIB_OUT=libsdmi_timing.so python stable-diffusion_amd/build.py
EXTRA = os.environ.get('SDMI_CXXFLAGS', '').split()
LIB = os.path.join(HERE, os.environ.get('SDMI_LIB_OUT', 'libsdmi.so'))
OBJ = os.path.join(HERE, 'build' if os.path.basename(LIB) == 'libsdmi.so' else 'build_' + os.path.splitext(os.path.basename(LIB))[0])
SOURCES = ['igemm_t1.hip', 'igemm_t0.hip', 'igemm_t2.hip', 'igemm.hip', 'conv3halo.hip', 'gemm_split16.hip', 'igemm5.hip', 'rowchain.hip', 'gnconv.hip', 'range.hip', 'attn.hip', 'attn_ctx.hip', 'norm.hip', 'small.hip', 'sampler.hip', 'unet.cpp', 'vae.cpp', 'clip.cpp', 'api.cpp', 'prof.cpp']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function',
         '-Wno-unused-variable']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (not os.path.isabs(c) or os.path.exists(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stamp():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), 'include')):
        for fn in sorted(os.listdir(root)):
            with open(os.path.join(root, fn), 'rb') as f:
                h.update(fn.encode())
                h.update(f.read())
    h.update(' '.join(FLAGS + EXTRA).encode())
    return h.hexdigest()


def is_current():
    stamp_file = os.path.join(OBJ, 'stamp')
    return os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == _stamp()


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    stamp_file = os.path.join(OBJ, 'stamp')
    if not force and is_current():
        return LIB
    hipcc = _hipcc()
    stamp_now = _stamp()       # of the sources as they are NOW: an edit made while the compilers run must leave the library stale

    inc_dirs = (CSRC, os.path.join(os.path.dirname(HERE), 'include'))

    def deps_hash(src):
        """hash of the source and of the project headers it includes (transitively; `#include "..."` only)"""
        import re
        seen, todo, h = set(), [os.path.join(CSRC, src)], hashlib.sha256()
        while todo:
            path = todo.pop()
            if path in seen:
                continue
            seen.add(path)
            with open(path, 'rb') as f:
                data = f.read()
            h.update(os.path.basename(path).encode())
            h.update(data)
            for inc in re.findall(rb'^\s*#\s*include\s*"([^"]+)"', data, re.M):
                for d in (os.path.dirname(path),) + inc_dirs:
                    cand = os.path.join(d, inc.decode())
                    if os.path.exists(cand):
                        todo.append(os.path.normpath(cand))
                        break
        return h.hexdigest()

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + '.o')
        extra = ['-ffp-contract=off'] if src == 'sampler.hip' else []    # bit-exact fp32 op order (see sampler.hip)
        cmd = [hipcc] + FLAGS + EXTRA + extra + (['-x', 'hip'] if src.endswith('.hip') else []) + ['-c', os.path.join(CSRC, src), '-o', obj]
        # per-object cache: igemm.hip alone takes minutes, so an object is rebuilt only when its source, a header or the flags changed
        key = hashlib.sha256((deps_hash(src) + ' '.join(cmd)).encode()).hexdigest()
        if not force and os.path.exists(obj) and os.path.exists(obj + '.key') and open(obj + '.key').read() == key:
            return obj
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(obj + '.key', 'w') as f:
            f.write(key)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    with open(stamp_file, 'w') as f:
        f.write(stamp_now)
    if verbose:
        print(f'built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
