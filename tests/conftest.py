import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'experiments: exercises a kernel that lost its same-box A/B and is only compiled with -DSDMI_EXPERIMENTS '
                            '(SDMI_CXXFLAGS=-DSDMI_EXPERIMENTS SDMI_LIB_OUT=libsdmi_exp.so python stable-diffusion_amd/build.py; '
                            'SDMI_LIB_PATH=.../libsdmi_exp.so pytest -m "gpu and experiments"); skipped against the product library')
    import torch
    # the GPU boxes expose 256 logical CPUs under a 16-CPU cgroup quota: torch's default thread count thrashes there
    torch.set_num_threads(min(32, _usable_cores()))
    # a clean checkout has no libsdmi.so (built artefacts are git-ignored): build it once (hipcc cross-compiles without a
    # GPU, ~1 min); a stale library is rebuilt too, so the tests never run against yesterday's kernels
    import importlib.util
    spec = importlib.util.spec_from_file_location('sdmi_build', os.path.join(ROOT, 'stable-diffusion_amd', 'build.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    if not b.is_current():
        b.build(force=False, verbose=False)


def pytest_collection_modifyitems(config, items):
    """GPU tests must not silently pass on a GPU-less host: they are skipped there unless selected with -m gpu,
    in which case a missing GPU is an error (the product path has no fallback)."""
    import torch
    if not torch.cuda.is_available():
        for item in items:
            if 'gpu' in item.keywords:
                item.add_marker(pytest.mark.skip(reason='no GPU on this host'))
        return
    from stable_diffusion_amd import _lib
    if not _lib.load().sdmi_has_experiments():
        for item in items:
            if 'experiments' in item.keywords:
                item.add_marker(pytest.mark.skip(reason='the product libsdmi.so is built without -DSDMI_EXPERIMENTS'))


@pytest.fixture(scope='session')
def lib():
    from stable_diffusion_amd import _lib
    return _lib.load()


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
