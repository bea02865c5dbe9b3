"""In-situ tuning of the implicit-GEMM tile / split-K choice on the MI355X (writes stable-diffusion_amd/tune_gfx950.txt).

    python tools/tune.py [--out PATH] [--rounds 64] [--reps 2] [--workloads unet64,unet96,...]

Every auto-configured GEMM launch of the real executors (UNet, first stage, text encoder) runs candidate
(round mod #candidates) of its shape while the library times it with HIP events on the launch stream
(sdmi_tune_begin / _round / _end, include/sdmi.h).  A candidate is therefore measured inside the real call: weights cold
in HBM (1.7 GB of them stream through per UNet call), activations warm from the producing kernel -- the conditions a
stand-alone sweep of one shape does not reproduce.  The resulting table is committed, so the choice (and with it the
split-K summation order, i.e. the exact output bits) is fixed.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'stable-diffusion_amd', 'tune_gfx950.txt'))
    ap.add_argument('--dump', default=None, help='also write the per-candidate timings here')
    ap.add_argument('--rounds', type=int, default=72)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--workloads', default='unet64,unet96,unet32,unet64b4,unet64b6,unet64b8,vaedec64,vaedec96,vaeenc512,clip')
    args = ap.parse_args()
    from stable_diffusion_amd import AutoencoderKLHIP, FrozenCLIPEmbedderHIP, UNetModelHIP, _lib
    from stable_diffusion_amd.synthetic import (SD_V1_UNET_KWARGS, SD_V1_VAE_DDCONFIG, randomize_, randomize_vae_,
                                                synthetic_clip_state_dict)
    lib = _lib.load()
    dev = torch.device('cuda')
    unet = UNetModelHIP(**SD_V1_UNET_KWARGS).to(dev).eval()
    randomize_(unet, 0)
    vae = AutoencoderKLHIP(SD_V1_VAE_DDCONFIG, None, 4).to(dev).eval()
    randomize_vae_(vae, 0)
    clip = None
    g = torch.Generator().manual_seed(0)

    def unet_fn(B, H):
        x = torch.randn(B, 4, H, H, generator=g).to(dev)
        ctx = (0.1 * torch.randn(B, 77, 768, generator=g)).to(dev)
        t = torch.full((B,), 481, device=dev)

        def fn():
            unet(x, t, context=ctx * 1.0)        # a fresh context object: the cross-attention K/V GEMMs run too
        return fn

    def vaedec_fn(H):
        z = torch.randn(1, 4, H, H, generator=g).to(dev)
        return lambda: vae.decode(z)

    def vaeenc_fn(S):
        img = torch.randn(1, 3, S, S, generator=g).to(dev)
        return lambda: vae.encode(img)

    def clip_fn():
        nonlocal clip
        if clip is None:
            clip = FrozenCLIPEmbedderHIP(tokenizer=object()).to(dev)
            clip.load_state_dict(synthetic_clip_state_dict(None, 0), strict=False)
            clip = clip.to(dev)
        ids = torch.randint(0, 49408, (2, 77), generator=g).to(dev)
        return lambda: clip.encode_ids(ids)

    makers = {
        'unet64': lambda: unet_fn(2, 64), 'unet96': lambda: unet_fn(2, 96), 'unet32': lambda: unet_fn(2, 32),
        'unet64b4': lambda: unet_fn(4, 64), 'unet64b6': lambda: unet_fn(6, 64), 'unet64b8': lambda: unet_fn(8, 64),
        'vaedec64': lambda: vaedec_fn(64), 'vaedec96': lambda: vaedec_fn(96), 'vaeenc512': lambda: vaeenc_fn(512),
        'clip': clip_fn,
    }
    fns = []
    for name in args.workloads.split(','):
        try:
            fn = makers[name]()
            fn()                                 # warm up outside the collection (packs weights, sizes workspaces)
            torch.cuda.synchronize()
            fns.append((name, fn))
        except Exception as e:                    # a workload that does not exist in this build is skipped, not fatal
            print(f'[tune] workload {name} skipped: {type(e).__name__}: {e}', flush=True)
    _lib.check(lib.sdmi_tune_begin())
    t0 = time.time()
    for rep in range(args.reps):
        for r in range(args.rounds):
            _lib.check(lib.sdmi_tune_round(r))
            for name, fn in fns:
                fn()
        torch.cuda.synchronize()
    n = C.c_int(0)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    _lib.check(lib.sdmi_tune_end(args.out.encode(), C.byref(n)))
    print(f'[tune] {n.value} shapes tuned over {args.reps} x {args.rounds} rounds of {[w for w, _ in fns]} in '
          f'{time.time() - t0:.1f} s -> {args.out}', flush=True)
    if args.dump:
        buf = C.create_string_buffer(8 << 20)
        _lib.check(lib.sdmi_tune_dump(buf, len(buf)))
        open(args.dump, 'w').write('# M N K ksize stride up mode splitk_req | tile splitk median_us min_us samples\n' + buf.value.decode())


if __name__ == '__main__':
    main()
