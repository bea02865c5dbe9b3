"""Import alias: the package directory is `stable-diffusion_amd/` (not an importable name), so
`import stable_diffusion_amd` loads it from there.  Yaml targets use this name, e.g.
`target: stable_diffusion_amd.unet.UNetModelHIP`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'stable-diffusion_amd')
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
