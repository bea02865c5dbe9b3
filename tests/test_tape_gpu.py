"""Launch tapes (csrc/tape.h, round 6): sdmi_unet_forward records the launch list of a (shape, workspace, timestep mode, knobs) once and replays
it with only the caller's pointers patched.  A replayed forward must give the bits the executor gives -- for other inputs, other output
tensors, other timesteps (table row, int64 tensor, float tensor), pinned and passed contexts, interleaved shapes, changed weights, flipped
knobs -- and SDMI_REPLAY_VERIFY=1 (the executor runs on every would-be replay and its launch list is compared with the patched tape) must pass."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.plan import SD_V1, TINY  # noqa: E402
from oracle.weights import make_inputs, make_state_dict  # noqa: E402


def _model(cfg, seed=0):
    from stable_diffusion_amd import UNetModelHIP
    m = UNetModelHIP(**cfg.ref_kwargs())
    m.load_state_dict(make_state_dict(cfg, seed), strict=True)
    return m.cuda().eval()


def _stats(m):
    a, b = C.c_int64(0), C.c_int64(0)
    from stable_diffusion_amd import _lib
    _lib.check(m._handle.lib.sdmi_unet_tape_stats(m._handle.h, C.byref(a), C.byref(b)))
    return a.value, b.value


def _calls(cfg):
    """(x, t, ctx, hint) of a little sampling-like sequence: three inputs, three timestep forms"""
    out = []
    for seed, ts in ((1, (981, 981)), (2, (961, 961)), (3, (1, 1))):
        x, t, ctx = make_inputs(cfg, 2, 16, 16, seed=seed, timesteps=ts)
        out.append((x.cuda(), t.cuda(), ctx.cuda(), ts[0]))
    return out


@pytest.mark.parametrize('cfg_name', ['tiny', 'sdv1'])
def test_replayed_forwards_equal_executed_forwards(cfg_name, monkeypatch):
    cfg = {'tiny': TINY, 'sdv1': SD_V1}[cfg_name]
    m = _model(cfg)
    calls = _calls(cfg)
    ctx = calls[0][2]

    def sequence():
        outs = []
        m.cache_timesteps([981, 961, 1])
        for x, t, c, hint in calls:                       # pinned context, hinted timesteps: the samplers' mode
            m.pin_context(ctx)
            m.hint_timestep(hint)
            outs.append(m(x, t, context=ctx).clone())
        m.unpin_context()
        m.cache_timesteps([])
        for x, t, c, hint in calls:                       # a fresh context every call, int64 timesteps computed by the call
            outs.append(m(x, t, context=c).clone())
        for x, t, c, hint in calls[:2]:                   # float timesteps (DPM-Solver)
            outs.append(m(x, t.float(), context=c).clone())
        x1, t1, c1 = make_inputs(cfg, 1, 8, 8, seed=9, ctx_len=40)       # another shape in between ...
        outs.append(m(x1.cuda(), t1.cuda(), context=c1.cuda()).clone())
        x, t, c, hint = calls[1]
        outs.append(m(x, t, context=c).clone())           # ... and back
        torch.cuda.synchronize()
        return outs

    monkeypatch.setenv('SDMI_REPLAY', '0')
    want = sequence()
    h0, r0 = _stats(m)
    assert h0 == 0 and r0 == 0
    monkeypatch.setenv('SDMI_REPLAY', '1')
    got1 = sequence()                                     # records (and already replays the repeats inside the sequence)
    got2 = sequence()                                     # replays
    h, r = _stats(m)
    print(f'[tape {cfg_name}] {h} forwards replayed, {r} recorded', flush=True)
    assert r >= 4 and h >= len(want) + 4
    for i, (w, a, b) in enumerate(zip(want, got1, got2)):
        assert torch.equal(w, a), (i, float((w - a).abs().max()))
        assert torch.equal(w, b), (i, float((w - b).abs().max()))
    monkeypatch.setenv('SDMI_REPLAY_VERIFY', '1')         # the executor runs; its launch list must equal the patched tape
    got3 = sequence()
    for i, (w, a) in enumerate(zip(want, got3)):
        assert torch.equal(w, a), i
    monkeypatch.delenv('SDMI_REPLAY_VERIFY')
    # a knob the library reads per call changes the tape's identity (here: the chain launches off -> other kernels, same bits)
    monkeypatch.setenv('SDMI_ST_HEAD', '0')
    x, t, c, hint = calls[0]
    e = m(x, t, context=c)
    assert torch.equal(e, want[3])
    monkeypatch.delenv('SDMI_ST_HEAD')
    # new weights: the tapes of the old packed buffers are dead
    m.load_state_dict(make_state_dict(cfg, 5), strict=True)
    e_new = m(x, t, context=c).clone()
    assert not torch.equal(e_new, want[3])
    monkeypatch.setenv('SDMI_REPLAY', '0')
    assert torch.equal(m(x, t, context=c), e_new)
