"""Host post-processing of the reference scripts (scripts/txt2img.py:313-327, scripts/img2img.py:264-280), SURVEY.md 8 f-4:

    x = model.decode_first_stage(samples)               -> AutoencoderKLHIP.decode_first_stage
    x = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)  \\
    x = 255. * x.permute(0, 2, 3, 1)  ... astype(uint8)  } `to_uint8_images`: one device pass, same bytes
    Image.fromarray(x).save(path)                       -> `save_png`
    put_watermark(img, wm_encoder)                      -> `put_watermark` (needs the optional `imwatermark` package)

The safety checker stays the reference's (a second CLIP model, out of scope: SURVEY.md 2 / DESIGN.md 7).
"""
import numpy as np
import torch

from . import _lib


@torch.no_grad()
def to_uint8_images(x_samples):
    """decode_first_stage output [B, C, H, W] (cuda, any float dtype) -> uint8 [B, H, W, C] (cuda), bit-identical to
    `(255. * clamp((x + 1) / 2, 0, 1).permute(0, 2, 3, 1).numpy()).astype(np.uint8)` of scripts/txt2img.py:314-324."""
    if not x_samples.is_cuda:
        raise RuntimeError('to_uint8_images runs on MI355X device tensors only (no CPU fallback)')
    x = x_samples.detach().float().contiguous()
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().sdmi_image_to_uint8(x.data_ptr(), out.data_ptr(), B, C, H, W, _lib.stream_ptr()))
    return out


def put_watermark(img, wm_encoder=None):
    """scripts/txt2img.py:41-46: dwtDct invisible watermark through `imwatermark`, when an encoder is given."""
    if wm_encoder is None:
        return img
    import cv2
    from PIL import Image
    arr = cv2.cvtColor(np.array(img), cv2.COLOR_RGB2BGR)
    arr = wm_encoder.encode(arr, 'dwtDct')
    return Image.fromarray(arr[:, :, ::-1])


def make_watermark_encoder(text='StableDiffusionV1'):
    """scripts/txt2img.py:261-264; returns None (no watermark) when `imwatermark` is not installed."""
    try:
        from imwatermark import WatermarkEncoder
    except ImportError:
        return None
    enc = WatermarkEncoder()
    enc.set_watermark('bytes', text.encode('utf-8'))
    return enc


def save_png(u8_hwc, path, wm_encoder=None):
    """One image (uint8 [H, W, C], any device) -> PNG file, as scripts/txt2img.py:324-326."""
    from PIL import Image
    arr = u8_hwc.cpu().numpy() if isinstance(u8_hwc, torch.Tensor) else np.asarray(u8_hwc)
    img = put_watermark(Image.fromarray(arr), wm_encoder)
    img.save(path)
    return path
