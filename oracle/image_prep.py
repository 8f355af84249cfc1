"""CPU restatement of the conditioning-image preparation that feeds the hot path (TEST INFRASTRUCTURE -- see
oracle/__init__.py): model/ctrl_helper.py:268-296 `prepare_images`, i.e. per frame diffusers'
VaeImageProcessor(do_convert_rgb=True, do_normalize=False).preprocess (model/ctrl_helper.py:56-58, :280) =
PIL convert("RGB") -> PIL resize((width, height), LANCZOS) -> uint8 / 255.0 -> NCHW float32, then the batch repeat (:284-286),
the leading clip axis (:288) and the CFG duplication (:291-294).

The only arithmetic is Pillow's 8-bit separable resampling (third-party dependency; src/libImaging/Resample.c of Pillow
12.x, installed in this image as PIL 12.2.0): integer work, restated here in numpy and PINNED bit for bit against PIL itself
in tests/test_image_prep.py (PIL is importable on the build box and on the GPU box)."""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2          # Resample.c: coefficients are 22-bit fixed point for 8-bit-per-channel images
LANCZOS_SUPPORT = 3.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def lanczos(x):
    return _sinc(x) * _sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c:precompute_coeffs + normalize_coeffs_8bpc for the whole-image box (in0 = 0, in1 = in_size):
    -> ksize, bounds int32 [out][2] (first source index, count), kk int32 [out][ksize] (22-bit fixed-point weights)"""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _resample_axis(img, out_size, axis):
    """one pass of ImagingResampleHorizontal_8bpc / Vertical_8bpc: img uint8 [..]; resamples `axis` to out_size"""
    in_size = img.shape[axis]
    _, bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)       # clip8: arithmetic shift, then clamp
    return np.moveaxis(out, 0, axis)


def pil_resize_lanczos(img_hwc, width, height):
    """PIL.Image.resize((width, height), LANCZOS) of an RGB uint8 image [H][W][3]: horizontal pass first, then vertical
    (ImagingResample); a pass is skipped when its size does not change"""
    out = img_hwc
    if out.shape[1] != width:
        out = _resample_axis(out, width, 1)
    if out.shape[0] != height:
        out = _resample_axis(out, height, 0)
    return out


def prepare_images(images_u8, width, height, batch_size, num_images_per_prompt, dtype=torch.float32,
                   do_classifier_free_guidance=False, guess_mode=False):
    """model/ctrl_helper.py:268-296 on RGB uint8 arrays [H][W][3] (what convert("RGB") yields): -> [1 | 2, F*rep, 3, height, width]"""
    frames = []
    for im in images_u8:
        r = pil_resize_lanczos(np.asarray(im, dtype=np.uint8), width, height)
        frames.append(torch.from_numpy(r.astype(np.float32) / 255.0).permute(2, 0, 1)[None])      # pil_to_numpy, numpy_to_pt
    x = torch.cat(frames, dim=0)
    x = x.repeat(batch_size * num_images_per_prompt, 1, 1, 1)
    x = x.unsqueeze(0).to(dtype)
    if do_classifier_free_guidance and not guess_mode:
        x = x.repeat(2, 1, 1, 1, 1)
    return x
