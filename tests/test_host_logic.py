"""CPU: host-side logic of the MI355X path -- the C ABI surface, the module/plugin mirror, sampler planning."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """libsdmi.so loads and exports exactly what include/sdmi.h declares (no compute calls: no GPU here)."""
    from stable_diffusion_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'sdmi.h')).read()
    declared = set(re.findall(r'\b(sdmi_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    nm = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (sdmi_[a-z0-9_]+)', nm))
    assert declared <= exported, declared - exported
    assert lib.sdmi_abi_version() == 17
    for name in declared:
        assert hasattr(lib, name)


def test_ctypes_struct_layout_matches_header():
    """sizeof / field order of the ABI structs as the C compiler sees them."""
    import ctypes as C
    from stable_diffusion_amd import _lib
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "sdmi.h"
int main(){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(sdmi_unet_cfg), offsetof(sdmi_unet_cfg, context_dim),
 sizeof(sdmi_igemm_desc), offsetof(sdmi_igemm_desc, w), offsetof(sdmi_igemm_desc, seg_dst), offsetof(sdmi_igemm_desc, dma),
 offsetof(sdmi_igemm_desc, asym_pad), sizeof(sdmi_vae_cfg), offsetof(sdmi_vae_cfg, embed_dim));
 printf("%zu %zu %zu %zu\n", sizeof(sdmi_clip_cfg), offsetof(sdmi_clip_cfg, max_positions), offsetof(sdmi_igemm_desc, pgn_out),
        offsetof(sdmi_igemm_desc, pgn_applied)); }
'''
    d = os.path.join(ROOT, 'stable-diffusion_amd', 'build')
    os.makedirs(d, exist_ok=True)
    c = os.path.join(d, 'abi_probe.c')
    open(c, 'w').write(src)
    exe = os.path.join(d, 'abi_probe')
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(_lib.UNetCfg), _lib.UNetCfg.context_dim.offset, C.sizeof(_lib.IGemmDesc), _lib.IGemmDesc.w.offset,
            _lib.IGemmDesc.seg_dst.offset, _lib.IGemmDesc.dma.offset, _lib.IGemmDesc.asym_pad.offset,
            C.sizeof(_lib.VaeCfg), _lib.VaeCfg.embed_dim.offset, C.sizeof(_lib.ClipCfg), _lib.ClipCfg.max_positions.offset,
            _lib.IGemmDesc.pgn_out.offset, _lib.IGemmDesc.pgn_applied.offset]
    assert got == want


def test_unet_shim_has_reference_parameter_names():
    """UNetModelHIP.state_dict() keys/shapes == the reference UNetModel's (oracle.weights is pinned to the reference by
    make_golden's strict load), so load_state_dict(sd, strict=False) at scripts/txt2img.py:56 fills every tensor."""
    from oracle.plan import SD_V1, TINY
    from oracle.weights import param_specs
    from stable_diffusion_amd import UNetModelHIP
    for cfg in (TINY,):
        m = UNetModelHIP(**cfg.ref_kwargs())
        mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        ref = {k: tuple(s) for k, s, _ in param_specs(cfg)}
        assert mine == ref
    # SD-v1 key list straight from the C library (no 3.4 GB allocation)
    from stable_diffusion_amd.unet import _Handle, make_cfg
    h = _Handle(make_cfg(4, 4, 320, 2, [1, 2, 4, 4], [4, 2, 1], 8, 1, 768))
    specs = h.weight_specs()
    assert {k: tuple(s) for k, s in specs} == {k: tuple(s) for k, s, _ in param_specs(SD_V1)}
    assert len(specs) == 686


def test_vae_shim_has_reference_parameter_names():
    """AutoencoderKLHIP.state_dict() keys/shapes == AutoencoderKL's (oracle.vae_ref.vae_param_specs is pinned to the
    reference Encoder / Decoder by make_golden_vae's strict load), so the first_stage_model.* part of a checkpoint loads."""
    from oracle.vae_ref import SD_VAE, TINY_VAE, vae_param_specs
    from stable_diffusion_amd import AutoencoderKLHIP
    m = AutoencoderKLHIP(TINY_VAE.ddconfig(), {'target': 'torch.nn.Identity'}, TINY_VAE.embed_dim)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == {k: tuple(s) for k, s, _ in vae_param_specs(TINY_VAE)}
    from stable_diffusion_amd.vae import _VaeHandle, make_vae_cfg
    for parts, enc, dec in ((1, False, True), (2, True, False), (3, True, True)):
        specs = _VaeHandle(make_vae_cfg(SD_VAE.ddconfig(), SD_VAE.embed_dim), parts).weight_specs()
        assert {k: tuple(s) for k, s in specs} == {k: tuple(s) for k, s, _ in vae_param_specs(SD_VAE, enc, dec)}
    with pytest.raises(RuntimeError, match='no CPU'):
        m.decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(NotImplementedError):
        AutoencoderKLHIP(dict(TINY_VAE.ddconfig(), attn_resolutions=[16]), None, 4)


def test_clip_shim_has_checkpoint_parameter_names():
    """FrozenCLIPEmbedderHIP.state_dict() == the `cond_stage_model.*` sub-tree of an SD-v1 checkpoint: transformers 4.19.2
    CLIPTextModel names (oracle.clip_ref.clip_param_specs, pinned to Hugging Face's model by make_golden_clip) + the
    persistent position_ids buffer."""
    from oracle import clip_ref
    from stable_diffusion_amd import FrozenCLIPEmbedderHIP
    tc = dict(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
              max_position_embeddings=77)
    m = FrozenCLIPEmbedderHIP(text_config=tc, tokenizer=object())
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref = {'transformer.' + k: tuple(s) for k, s, _ in clip_ref.clip_param_specs(clip_ref.TINY_CLIP)}
    ref['transformer.text_model.embeddings.position_ids'] = (1, 77)
    assert mine == ref
    from stable_diffusion_amd.clip import CLIP_VIT_L14_TEXT, _ClipHandle, make_clip_cfg
    specs = _ClipHandle(make_clip_cfg(CLIP_VIT_L14_TEXT)).weight_specs()
    assert {k: tuple(s) for k, s in specs} == {k: tuple(s) for k, s, _ in clip_ref.clip_param_specs(clip_ref.SD_CLIP)}
    assert len(specs) == 196
    with pytest.raises(RuntimeError, match='no CPU'):
        m.encode_ids(torch.zeros(1, 77, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        FrozenCLIPEmbedderHIP(text_config=dict(tc, hidden_act='gelu'), tokenizer=object())


def test_shim_refuses_cpu_tensors_and_foreign_configs():
    from oracle.plan import TINY
    from stable_diffusion_amd import UNetModelHIP
    m = UNetModelHIP(**TINY.ref_kwargs())
    with pytest.raises(RuntimeError, match='no CPU'):
        m(torch.zeros(1, 4, 8, 8), torch.zeros(1, dtype=torch.long), context=torch.zeros(1, 77, TINY.context_dim))
    with pytest.raises(NotImplementedError):
        UNetModelHIP(image_size=32, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=2,
                     attention_resolutions=[1])
    from stable_diffusion_amd import _lib
    with pytest.raises(_lib.SdmiError):
        _lib.check(_lib.load().sdmi_unet_finalize(m._handle.h))       # weights never set -> error string, not a crash
    assert b'weight not set' in _lib.load().sdmi_last_error()


def test_workspace_query_is_pure_host_logic():
    from stable_diffusion_amd.unet import _Handle, make_cfg
    h = _Handle(make_cfg(4, 4, 320, 2, [1, 2, 4, 4], [4, 2, 1], 8, 1, 768))
    b64 = h.lib.sdmi_unet_workspace_bytes(h.h, 2, 64, 64, 77)
    b32 = h.lib.sdmi_unet_workspace_bytes(h.h, 2, 32, 32, 77)
    assert 50e6 < b32 < b64 < 2e9
    assert h.lib.sdmi_unet_workspace_bytes(h.h, 2, 63, 64, 77) == 0   # not divisible by 8
    assert b'divisible' in h.lib.sdmi_last_error()


def test_sampler_tables_and_plan_match_oracle():
    from oracle import samplers_ref
    from stable_diffusion_amd import samplers
    _, ac = samplers_ref.make_alphas_cumprod()
    for S in (50, 10, 30, 7):
        ts = samplers.make_ddim_timesteps(S, 1000)
        assert np.array_equal(ts, samplers_ref.make_ddim_timesteps(S))
        a, b = samplers.make_tables(ac, ts, 0.0), samplers_ref.make_sampling_tables(ac, ts, 0.0)
        for k in a:
            assert np.array_equal(a[k], np.asarray(b[k], dtype=np.float32)), k
    plan = samplers.plms_plan(samplers.make_ddim_timesteps(50, 1000))
    assert len(plan) == 50
    assert plan[0] == (0, 49, 981, 961, 4) and plan[1][4] == 1 and plan[2][4] == 2 and plan[3][4] == 3 and plan[10][4] == 3
    assert plan[-1][:4] == (49, 0, 1, 1)        # last step: t_next == t (plms.py:145)
    with pytest.raises(ValueError):
        samplers.PLMSSamplerHIP(type('M', (), {'num_timesteps': 1000})()).make_schedule(10, ddim_eta=0.5)


def test_dpm_solver_plan_reproduces_the_reference(golden_dir):
    """The host-side DPM-Solver++ plan (schedule interpolation, step orders incl. lower_order_final, coefficients) driven
    through a torch emulation of sdmi_dpm_solver_step's arithmetic reproduces the reference DPMSolverSampler's end points
    exactly (same fp32 ops in the same order on the same CPU)."""
    from stable_diffusion_amd.samplers import _DiscreteVP, dpm_plan
    D = np.load(os.path.join(golden_dir, 'dpm_solver.npz'))
    ac = torch.from_numpy(D['alphas_cumprod'])
    x_T, c, uc = (torch.from_numpy(D[k]) for k in ('x_T', 'c', 'uc'))

    def stub(x, t, cc):
        return torch.tanh(0.7 * x + 0.001 * t.float()[:, None, None, None]) * 0.9 + 0.05 * cc.mean(dim=(1, 2))[:, None, None, None]
    for S, scale, ucond in ((20, 7.5, uc), (10, 7.5, uc), (50, 5.0, uc), (12, 1.0, None)):
        plan = dpm_plan(_DiscreteVP(ac), S)
        assert [p[3] for p in plan] == [1] + [2] * (S - 2) + [1 if S < 15 else 2]
        x, m_prev = x_T.clone(), None
        for (t_in, al, sg, order, cx, a, ir) in plan:
            tt = torch.full((2,), t_in)
            if ucond is None:
                e = stub(x, tt, c)
            else:
                eu, ec = stub(torch.cat([x] * 2), torch.cat([tt] * 2), torch.cat([ucond, c])).chunk(2)
                e = eu + scale * (ec - eu)
            m0 = (x - sg * e) / al
            xt = cx * x - a * m0
            if order == 2:
                xt = xt - (0.5 * a) * (ir * (m0 - m_prev))
            m_prev, x = m0, xt
        assert torch.equal(x, torch.from_numpy(D[f'dpm_{S}_{scale}']))
    with pytest.raises(ValueError):
        dpm_plan(_DiscreteVP(ac), 1)


def test_ldm_shim_schedule_equals_oracle():
    from oracle import samplers_ref
    from stable_diffusion_amd.ldm_shim import LatentDiffusionHIP
    ld = LatentDiffusionHIP(torch.nn.Identity())
    betas, ac = samplers_ref.make_alphas_cumprod()
    assert np.array_equal(ld.alphas_cumprod.numpy(), ac) and np.array_equal(ld.betas.numpy(), betas)
    assert ld.num_timesteps == 1000 and float(ld.alphas_cumprod_prev[0]) == 1.0


@pytest.mark.skipif(not os.path.isdir('/root/reference/ldm'), reason='reference checkout only exists in the build container')
def test_plugin_seam_against_the_real_reference():
    """The reference's own plugin mechanism (`instantiate_from_config`, ldm/util.py:78-93) resolves the patched
    `unet_config.target` to UNetModelHIP with the yaml's params, and the resulting state_dict keys / shapes equal the
    real reference UNetModel's (built on the meta device), so `load_state_dict(sd, strict=False)` fills everything."""
    import sys
    import yaml
    sys.path.insert(0, '/root/reference')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import run_reference_script as launcher
    launcher.install_stubs(have_gpu=True, offline_stubs=True)   # stand-ins only; nn.Module.cuda stays untouched
    from ldm.util import instantiate_from_config
    cfg = yaml.safe_load(open('/root/reference/configs/stable-diffusion/v1-inference.yaml'))
    unet_cfg = cfg['model']['params']['unet_config']
    assert unet_cfg['target'] == 'ldm.modules.diffusionmodules.openaimodel.UNetModel'
    with torch.device('meta'):
        ref = instantiate_from_config(unet_cfg)
    ref_keys = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    unet_cfg = dict(unet_cfg, target='stable_diffusion_amd.unet.UNetModelHIP')
    with torch.device('meta'):
        mine = instantiate_from_config(unet_cfg)
    from stable_diffusion_amd import UNetModelHIP
    assert isinstance(mine, UNetModelHIP)
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == ref_keys
    assert len(ref_keys) == 686
    # first stage: same mechanism (LatentDiffusion.instantiate_first_stage, ddpm.py:502-507); the reference class itself
    # needs pytorch_lightning + taming (stand-ins installed above), its Encoder / Decoder are the real ones
    fs_cfg = cfg['model']['params']['first_stage_config']
    assert fs_cfg['target'] == 'ldm.models.autoencoder.AutoencoderKL'
    import contextlib
    import io
    with torch.device('meta'), contextlib.redirect_stdout(io.StringIO()):
        ref_vae = instantiate_from_config(fs_cfg)
    ref_keys = {k: tuple(v.shape) for k, v in ref_vae.state_dict().items() if not k.startswith('loss.')}
    with torch.device('meta'):
        mine = instantiate_from_config(dict(fs_cfg, target='stable_diffusion_amd.vae.AutoencoderKLHIP'))
    from stable_diffusion_amd import AutoencoderKLHIP
    assert isinstance(mine, AutoencoderKLHIP)
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == ref_keys
    assert len(ref_keys) == 248
    # cond stage: `instantiate_from_config(cond_stage_config)` (ddpm.py:509-520); the yaml passes no params, the class
    # defaults are the reference's (version / device / max_length, modules.py:139)
    cs_cfg = cfg['model']['params']['cond_stage_config']
    assert cs_cfg['target'] == 'ldm.modules.encoders.modules.FrozenCLIPEmbedder'
    with torch.device('meta'):
        mine = instantiate_from_config(dict(cs_cfg, target='stable_diffusion_amd.clip.FrozenCLIPEmbedderHIP'))
    from oracle import clip_ref
    from stable_diffusion_amd import FrozenCLIPEmbedderHIP
    assert isinstance(mine, FrozenCLIPEmbedderHIP) and mine.max_length == 77 and mine.device == 'cuda'
    want = {'transformer.' + k: tuple(s) for k, s, _ in clip_ref.clip_param_specs(clip_ref.SD_CLIP)}
    want['transformer.text_model.embeddings.position_ids'] = (1, 77)
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == want
    assert hasattr(mine, 'encode') and hasattr(mine, 'freeze')


@pytest.mark.parametrize('name', ['bench_r01.json', 'bench_r02.json'])
def test_committed_bench_line_has_the_contract_keys(name):
    """profiles/bench_rNN.json is the JSON line `python bench.py` printed on the MI355X: the driver's contract keys, the
    roofline / cpu_baseline objects, and internally consistent numbers."""
    import json
    d = json.load(open(os.path.join(ROOT, 'profiles', name)))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'images/s' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['n_gpus'] == 1
    assert d['vs_baseline'] is None and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - d['n_gpus'] * 1000.0 / d['ms_per_step']) < 1e-6 * d['value'] + 1e-9
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['unit'] == d['unit'] and d['value'] / c['value'] >= 8.0   # north_star: >= 8x


def test_cpu_baseline_kind_follows_the_visibility_of_the_reference(monkeypatch):
    """bench.py's CPU comparator is the reference's own UNetModel where /root/reference is visible (kind "reference": the
    build container, recorded in profiles/cpu_baseline_reference_r02.json) and the oracle port where it is not (a GPU
    box: nothing under bench.py may read /root/reference there)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('sdmi_bench', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv('SD_REFERENCE', '/nonexistent/reference')
    bundle = os.path.join(ROOT, 'oracle', '_ref', 'refbundle', 'ldm', 'modules', 'diffusionmodules', 'openaimodel.pyc')
    real_exists = os.path.exists
    monkeypatch.setattr(os.path, 'exists', lambda p_: False if p_ == bundle else real_exists(p_))
    assert bench.reference_unet() == (None, None)       # neither the sources nor the bytecode bundle: the oracle port is timed
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'cpu_baseline_reference_r02.json')))['cpu_baseline']
    assert rec['kind'] == 'reference' and rec['unit'] == 'images/s' and 0 < rec['value'] < 0.1


def test_gpu_side_never_reads_the_reference_tree():
    """/root/reference does not exist on a GPU box: the `-m gpu` tests and smoke() must not mention it, and bench.py may
    only probe it behind an existence check (the CPU comparator's `kind`)."""
    import glob
    for f in glob.glob(os.path.join(ROOT, 'tests', 'test_*_gpu.py')) + [os.path.join(ROOT, 'tests', 'kernels.py')]:
        text = open(f).read()             # (docstrings may mention the path; code must not use it or import from it)
        for needle in ("'/root/reference", '"/root/reference', 'SD_REFERENCE', 'import ldm', 'from ldm'):
            assert needle not in text, (f, needle)
    entry = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    smoke = entry[entry.index('def smoke'):]
    assert '/root/reference' not in smoke and 'SD_REFERENCE' not in smoke
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    body = bench[bench.index('def reference_unet'):bench.index('def cpu_baseline')]
    assert 'os.path.isdir' in body and bench.count("'/root/reference'") == body.count("'/root/reference'") == 1


def test_committed_tuning_table_is_well_formed():
    """stable-diffusion_amd/tune_gfx950.txt fixes the (tile, split-K) choice per GEMM shape -- and with the split the summation
    order, i.e. the exact output bits: every line must parse, name an existing tile and a split the kernels support, and use
    the halo-staged conv tiles (14..17) only for stride-1 3x3 convolutions."""
    path = os.path.join(ROOT, 'stable-diffusion_amd', 'tune_gfx950.txt')
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith('#')]
    assert len(rows) >= 300
    seen = set()
    for r in rows:
        assert len(r) == 11, r
        M, N, K, ksize, stride, up, mode, req, tile, splitk = (int(x) for x in r[:10])
        us = float(r[10])
        # key ksize: 1 / 3 = the generic kernel, 11 = the split-fp16 dense GEMM family (gemm_split16.hip), 13 = the GroupNorm-folding
        # halo conv (conv3halo_gn_kernel; tile 99 = "run it as two launches")
        assert M > 0 and N > 0 and K > 0 and K % 64 == 0 and ksize in (1, 3, 11, 13) and stride in (1, 2) and up in (0, 1) and mode in (0, 1, 2)
        assert (0 <= tile < 23 or (tile == 99 and ksize == 13)) and 1 <= splitk <= 16 and us > 0
        if ksize == 11:
            assert tile in (0, 1, 2, 4, 5, 8, 10) and mode == 0, r       # the tiles gemm_split16.hip instantiates
        if ksize == 13:
            assert tile in (14, 15, 16, 17, 99) and stride == 1 and up == 0, r
        if 14 <= tile <= 17:
            assert ksize in (3, 13) and stride == 1 and up == 0, r
        if mode == 1:
            assert splitk == 1, r          # the GEGLU epilogue pairs columns inside a tile: never split
        key = tuple(r[:8])
        assert key not in seen, key
        seen.add(key)


def test_package_modules_reference_no_undefined_module_names():
    """Every `name.attr` in the host package resolves to an import, a local definition or a builtin (a missing `import os` in a
    GPU-only code path is invisible to the CPU suite otherwise)."""
    import ast
    import builtins
    pkg = os.path.join(ROOT, 'stable-diffusion_amd')
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith('.py'):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        known = set(dir(builtins))
        for n in ast.walk(tree):
            if isinstance(n, ast.Import):
                known |= {(a.asname or a.name).split('.')[0] for a in n.names}
            elif isinstance(n, ast.ImportFrom):
                known |= {a.asname or a.name for a in n.names}
            elif isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
                known.add(n.name)
            elif isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                known.add(n.id)
            elif isinstance(n, ast.arguments):
                known |= {a.arg for a in n.args + n.kwonlyargs + n.posonlyargs}
                known |= {a.arg for a in (n.vararg, n.kwarg) if a is not None}
            elif isinstance(n, ast.ExceptHandler) and n.name:
                known.add(n.name)
        used = {n.value.id for n in ast.walk(tree) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name)}
        assert used <= known, (fn, sorted(used - known))


def test_timestep_table_host_plumbing(monkeypatch):
    """UNetModelHIP.cache_timesteps / hint_timestep / forward as far as the C calls (recorded by a stand-in library): the list
    is deduplicated and sorted, the hint is one-shot, reaches every chunk of a > 8-row batch, and only integer timesteps."""
    from stable_diffusion_amd import _lib
    from stable_diffusion_amd.unet import UNetModelHIP
    calls = []

    class FakeLib:
        def sdmi_unet_cache_timesteps(self, h, arr, n, s):
            calls.append(('cache', list(arr)[:n])); return 0

        def sdmi_unet_hint_timestep(self, h, t):
            calls.append(('hint', t)); return 0

        def sdmi_unet_forward(self, *a):
            calls.append(('forward', a[6])); return 0          # a[6] = B

        def sdmi_unet_workspace_bytes(self, h, B, H, W, L):
            return 256

    class Handle:
        lib = FakeLib(); h = None

    class FakeTensor(torch.Tensor):
        pass

    m = UNetModelHIP.__new__(UNetModelHIP)
    torch.nn.Module.__init__(m)
    m._handle = Handle(); m._needs_pack = lambda: False
    m.in_channels, m.out_channels, m.context_dim = 4, 4, 8
    m._ws = None; m._pinned = None; m._pinned_ctx = None; m._ctx_ref = None; m._ctx_ver = None; m._ctx_shape = None
    monkeypatch.setattr(_lib, 'stream_ptr', lambda: None)
    m.cache_timesteps([981, 5, 5, 3])
    assert calls == [('cache', [3, 5, 981])]
    monkeypatch.setenv('SDMI_T_TABLE', '0')
    m.cache_timesteps([1, 2])
    assert len(calls) == 1
    # forward(): CPU tensors are refused before any C call, so drive _forward_rows' hint logic through a patched is_cuda check
    x = torch.zeros(10, 4, 8, 8); ctx = torch.zeros(10, 77, 8)
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    m.hint_timestep(5)
    m.forward(x, torch.full((10,), 5, dtype=torch.long), context=ctx)
    assert [c for c in calls[1:]] == [('hint', 5), ('forward', 8), ('hint', 5), ('forward', 2)]
    del calls[:]
    m.forward(x[:2], torch.full((2,), 5, dtype=torch.long), context=ctx[:2])           # the hint was consumed
    assert calls == [('forward', 2)]
    del calls[:]
    m.hint_timestep(5)
    m.forward(x[:2], torch.full((2,), 5.0), context=ctx[:2])                           # float timesteps (DPM-Solver): no hint
    assert calls == [('forward', 2)]
    monkeypatch.setenv('SDMI_CHECK_T_HINT', '1')                                       # debug: a wrong hint is an error
    m.hint_timestep(7)
    with pytest.raises(RuntimeError, match='hint_timestep'):
        m.forward(x[:2], torch.full((2,), 5, dtype=torch.long), context=ctx[:2])


def test_design_knob_table_matches_the_sources():
    """Every SDMI_* environment knob that DESIGN.md's table documents is read somewhere in the sources, and every knob the library /
    host package reads is documented (or a profiles / tools file for the bisecting-only ones) -- the table cannot rot."""
    import re
    design = open(os.path.join(ROOT, 'DESIGN.md')).read()
    table = design[design.index('### Environment knobs'):design.index('### What the measurements say (MI355X, round 1)')]
    documented = set(re.findall(r'SDMI_[A-Z0-9_]+', table))
    read = set()
    for base, exts in ((os.path.join(ROOT, 'stable-diffusion_amd', 'csrc'), ('.hip', '.cpp', '.h')),
                       (os.path.join(ROOT, 'stable-diffusion_amd'), ('.py',)), (ROOT, ('bench.py',))):
        for fn in sorted(os.listdir(base)):
            if fn.endswith(exts):
                src = open(os.path.join(base, fn)).read()
                read |= set(re.findall(r'(?:getenv|env_int|SDMI_EXP_ENV|environ\.get|environ\[)\(?\s*[\'"](SDMI_[A-Z0-9_]+)', src))
    # build-time / test-only names that are not run-time knobs of the library
    not_knobs = {'SDMI_CXXFLAGS', 'SDMI_LIB_OUT', 'SDMI_EXPERIMENTS', 'SDMI_REGEN_GOLDEN', 'SDMI_IGEMM_TIMING', 'SDMI_EPI_ABL', 'SDMI_ATTN_NW', 'SDMI_ATTN_ABL',
                 'SDMI_NT_STORES', 'SDMI_GN_VISIBLE', 'SDMI_GN_XBAR', 'SDMI_GN_POISON', 'SDMI_LIB_PATH'}
    missing_in_code = sorted(k for k in documented - read - not_knobs)
    assert not missing_in_code, f'documented in DESIGN.md but read nowhere: {missing_in_code}'
    undocumented = sorted(k for k in read - documented - not_knobs)
    assert not undocumented, f'read by the sources but missing from the DESIGN.md knob table: {undocumented}'


def test_reference_bytecode_bundle_builds_and_imports_sourceless(tmp_path):
    """oracle/build_ref_bundle.py (the recipe behind tests/test_reference_script_gpu.py): compiles the reference's modules where
    they lie into sourceless .pyc files -- no source text in the bundle --, the manifest pins the source hashes, and a fresh
    interpreter imports `ldm.util.instantiate_from_config` from the bundle alone."""
    import hashlib
    import json
    ref = '/root/reference'
    if not os.path.isdir(os.path.join(ref, 'ldm')):
        pytest.skip('no reference checkout here (the bundle is built where /root/reference is visible)')
    import importlib.util
    spec = importlib.util.spec_from_file_location('sdmi_ref_bundle_t', os.path.join(ROOT, 'oracle', 'build_ref_bundle.py'))
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    out = rb.build(ref, str(tmp_path / 'bundle'), verbose=False)
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs]
    assert files and not [f for f in files if f.endswith(('.py', '.yaml'))]          # bytecode + two JSON files only
    man = json.load(open(os.path.join(out, 'MANIFEST.json')))
    assert 'scripts/txt2img.py' in man['files'] and 'ldm/models/diffusion/plms.py' in man['files']
    for rel, h in list(man['files'].items())[:5]:
        assert hashlib.sha256(open(os.path.join(ref, rel), 'rb').read()).hexdigest() == h
    cfg = json.load(open(os.path.join(out, 'v1-inference.json')))
    assert cfg['model']['params']['unet_config']['target'] == 'ldm.modules.diffusionmodules.openaimodel.UNetModel'
    code = ("import sys; sys.path.insert(0, sys.argv[1]); import ldm.util as u; "
            "assert u.__file__.endswith('.pyc'), u.__file__; print(u.instantiate_from_config.__name__)")
    r = subprocess.run([sys.executable, '-c', code, out], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and 'instantiate_from_config' in r.stdout, r.stderr[-800:]


def test_register_order_slab_layout_model():
    """The split-K slab layout of round 4 (csrc/igemm_dev.h igemm_epilogue: slab_tiled branch; csrc/igemm.hip tiled_quad + quad_transpose),
    restated in Python: the GEMM side's (row, column) -> slab float index and the reduction side's thread -> (row, 4 columns) decode must
    be inverse to each other for every tile geometry the launchers use -- a host-side guard for edits to either half."""
    def gemm_index(m, n, BM, BN, WM, WN, tiles_n):
        """float index inside one split's slab of accumulator element (m, n): MFMA 32x32 C layout, 16 bytes = 4 consecutive rows of a lane"""
        WTM, WTN, NT = BM // WM, BN // WN, WM * WN * 64
        TN = WTN // 32
        tile_m, mm = divmod(m, BM); tile_n, nn = divmod(n, BN)
        wm, mw = divmod(mm, WTM); wn, nw = divmod(nn, WTN)
        i, r32 = divmod(mw, 32); j, l31 = divmod(nw, 32)
        # row inside a 32x32 tile = (r & 3) + 8 * (r >> 2) + 4 * lg
        e, lg, r4 = r32 & 3, (r32 >> 2) & 1, r32 >> 3
        tid = (wm * WN + wn) * 64 + lg * 32 + l31
        return (tile_m * tiles_n + tile_n) * BM * BN + ((i * TN + j) * 4 + r4) * (NT * 4) + tid * 4 + e

    def reduce_decode(q, lane, BM, BN, WM, WN, tiles_n):
        """splitk_reduce_tiled_kernel: thread q (slab quad index) owns, after the lane transpose, row m and columns n .. n + 3"""
        WTM, WTN, NT = BM // WM, BN // WN, WM * WN * 64
        TN = WTN // 32
        qpt = BM * BN // 4
        tile, rem = divmod(q, qpt)
        blk, t_in = divmod(rem, NT)
        assert t_in % 64 == lane
        wave = t_in >> 6
        ij, r4 = blk >> 2, blk & 3
        i, j = divmod(ij, TN)
        wm, wn = divmod(wave, WN)
        tile_m, tile_n = divmod(tile, tiles_n)
        l31, lg = lane & 31, lane >> 5
        m = tile_m * BM + wm * WTM + i * 32 + 8 * r4 + 4 * lg + (l31 & 3)
        n = tile_n * BN + wn * WTN + j * 32 + (l31 & ~3)
        return m, n

    geoms = [(64, 64, 2, 2), (128, 64, 2, 2), (128, 128, 2, 2), (256, 128, 4, 2), (64, 128, 2, 2), (128, 128, 4, 2), (128, 256, 2, 4),
             (64, 256, 1, 4), (256, 64, 4, 1), (256, 64, 4, 2), (128, 64, 2, 2)]
    for BM, BN, WM, WN in geoms:
        tiles_m, tiles_n = 2, 3
        M, N = tiles_m * BM, tiles_n * BN
        owner = {}
        nquads = tiles_m * tiles_n * BM * BN // 4
        for q in range(0, nquads, 7):                       # a stride that visits every lane / wave / block position
            lane = q % 64
            # the slab quad thread q LOADS holds rows 8 r4 + 4 lg + e (e = 0..3) of column l31: its four floats
            m, n = reduce_decode(q, lane, BM, BN, WM, WN, tiles_n)
            # after the transpose it owns (m, n..n+3): those four elements must sit at element (m & 3) of the quads of the four lanes of its lane quad
            for c in range(4):
                idx = gemm_index(m, n + c, BM, BN, WM, WN, tiles_n)
                src_q = idx // 4
                assert idx % 4 == (m & 3)
                assert src_q // 64 == q // 64 and (src_q % 64) // 4 == lane // 4 and (src_q % 64) % 4 == c, (BM, BN, q, c)
            owner[(m, n)] = q
        assert all(0 <= m < M and 0 <= n < N and n % 4 == 0 for m, n in owner)
        # and the GEMM-side index is a bijection onto [0, M * N)
        seen = set()
        for m in range(0, M, 5):
            for n in range(0, N, 3):
                seen.add(gemm_index(m, n, BM, BN, WM, WN, tiles_n))
        assert len(seen) == len(range(0, M, 5)) * len(range(0, N, 3)) and max(seen) < M * N


def test_row_to_sample_magic_division_model():
    """csrc/igemm_dev.h fast_div_hw / div_magic_hw (the implicit GEMM's row -> sample split m / (Hout*Wout)): the 2^48 multiply-shift
    form is exact for every row index of every supported batch, and the 2^40 form it replaced was NOT at four 768 x 768 first-stage
    maps (m * d >= 2^40: the last pixel of the last sample came out as sample 4 -- tests/test_vae_gpu.py::test_vae_decode_outputs_beyond_2_gb)."""
    def magic(d, s):
        return ((1 << s) + d - 1) // d

    def fdiv(m, mg, s):
        prod = m * mg
        assert prod < (1 << 64), 'the product must fit the 64-bit multiply'
        return prod >> s
    for hw in (64, 77, 256, 1024, 4096, 9216, 96 * 96 * 64, 768 * 768, 1024 * 1024):
        for B in (1, 2, 3, 4, 8):
            M = B * hw
            if M >= 1 << 31:
                continue
            mg = magic(hw, 48)
            ms = set()
            for b in range(B):
                ms.update((b * hw, b * hw + 1, b * hw + hw // 2, (b + 1) * hw - 2, (b + 1) * hw - 1))
            for m in ms:
                assert fdiv(m, mg, 48) == m // hw, (hw, B, m)
    hw, m = 768 * 768, 4 * 768 * 768 - 1
    assert fdiv(m, magic(hw, 40), 40) == 4 and m // hw == 3       # the bug the 2^48 form fixes


def test_prefetch_wave_covers_every_line_of_a_unit_exactly_once():
    """CPU model of the XCD-cooperative prefetch wave of the row-strip chain kernels (csrc/rowchain.hip rc_pf_row, csrc/gnconv.hip): the
    workgroups of an XCD (slot = blockIdx / 8 of gridDim / 8) share the 160 weight rows of a 20 KB unit -- lane -> line slot + (lane / 4) * nslots,
    32-byte sector lane % 4.  For every grid the chain kernels are launched with (M / 32 workgroups, M = B * ntok), each of the 160 lines x 4
    sectors is touched exactly once per XCD, and the row offset is the loaders' (64 w + [0, 32) of compute wave w)."""
    for M in (64 * 32 // 2, 8192, 4096, 18432, 16384, 2048, 256 * 32):
        ntiles = M // 32
        if ntiles % 8:
            continue                                   # (no prefetch: the kernel's slot is out of range)
        nslots = max(ntiles // 8, 10)
        seen = {}
        for slot in range(ntiles // 8):
            for lane in range(64):
                idx = slot + (lane >> 2) * nslots
                if slot < nslots and idx < 160:
                    row = 64 * (idx >> 5) + (idx & 31)
                    key = (row, lane & 3)
                    assert key not in seen, (M, slot, lane)
                    seen[key] = 1
        rows = sorted({r for r, _ in seen})
        want = sorted(64 * w + r for w in range(5) for r in range(32))
        if ntiles // 8 >= 10:                          # enough workgroups per XCD: full coverage
            assert rows == want and len(seen) == 160 * 4, (M, len(seen))
        else:                                          # tiny grids: a prefix of the slots exists; whatever is touched is a valid line
            assert set(rows) <= set(want)
