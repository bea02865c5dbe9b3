"""Checkpoint -> packed-weight blob for libsdmi.

    python tools/pack_checkpoint.py --ckpt sd-v1-4.ckpt --out unet.sdmi          # needs the MI355X (the packers are kernels)
    python tools/pack_checkpoint.py --ckpt synthetic:0 --out unet.sdmi

Reads the `model.diffusion_model.*` tensors of a Stable Diffusion v1 checkpoint (what `load_model_from_config` hands to
`load_state_dict`, scripts/txt2img.py:49-66), lets the library repack them once (fp16 [N][K] chunk-major conv weights,
split-fp16 1x1 weights, interleaved GEGLU rows, ...) and writes the packed buffers behind a header that pins the UNet
configuration and the ABI version.  `UNetModelHIP.load_packed(path)` then starts from the 1.9 GB blob (mmap + one
host->device copy per buffer) instead of the 4 GB fp32 pickle + repack.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--ckpt', required=True, help='path to an SD-v1 checkpoint, or synthetic[:seed]')
    ap.add_argument('--out', required=True)
    args = ap.parse_args()
    from stable_diffusion_amd import UNetModelHIP
    from stable_diffusion_amd.synthetic import SD_V1_UNET_KWARGS, synthetic_state_dict
    t0 = time.perf_counter()
    if args.ckpt.startswith('synthetic'):
        seed = int(args.ckpt.split(':')[1]) if ':' in args.ckpt else 0
        sd = synthetic_state_dict(SD_V1_UNET_KWARGS, seed)
    else:
        full = torch.load(args.ckpt, map_location='cpu')
        full = full.get('state_dict', full)
        pre = 'model.diffusion_model.'
        sd = {k[len(pre):]: v for k, v in full.items() if k.startswith(pre)}
    unet = UNetModelHIP(**SD_V1_UNET_KWARGS)
    missing, unexpected = unet.load_state_dict(sd, strict=False)
    if missing:
        raise SystemExit(f'checkpoint lacks {len(missing)} UNet tensors, e.g. {missing[:3]}')
    unet = unet.cuda()
    n = unet.save_packed(args.out)
    print(f'wrote {args.out}: {n / 1e9:.2f} GB packed UNet weights in {time.perf_counter() - t0:.1f} s '
          f'({len(unexpected)} non-UNet keys ignored)')


if __name__ == '__main__':
    main()
