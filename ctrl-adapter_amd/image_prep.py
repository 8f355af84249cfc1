"""prepare_images -- mirror of the reference's ControlNetHelper.prepare_images (model/ctrl_helper.py:268-296) on libctrlhip.

Per frame the reference runs diffusers' VaeImageProcessor(do_convert_rgb=True, do_normalize=False).preprocess
(model/ctrl_helper.py:56-58,:280): PIL convert("RGB") -> PIL resize((width, height), LANCZOS) -> uint8 / 255 -> NCHW
float32; then repeats the frames batch_size * num_images_per_prompt times (:284-286), adds the leading axis (:288), casts
to `dtype` and duplicates for classifier-free guidance (:291-294).  Here the uint8 frames go to the GPU once and two HIP
kernels (csrc/imageprep.hip) do Pillow's 8-bit separable Lanczos resampling -- bit-exact, it is integer arithmetic --
the /255, the layout change and every replica in one sweep.  Same signature and return value as the reference method.
The 22-bit fixed-point weight tables of Pillow's precompute_coeffs are computed on the host in double precision and cached
per (input size, output size).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L

_PRECISION_BITS = 32 - 8 - 2       # Pillow Resample.c, 8 bits per channel
_tables = {}


def _lanczos(x):
    if not (-3.0 <= x < 3.0):
        return 0.0

    def sinc(v):
        if v == 0.0:
            return 1.0
        v *= math.pi
        return math.sin(v) / v
    return sinc(x) * sinc(x / 3.0)


def _coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (box = the whole image) -> (ksize, bounds [out][2], kk [out][ksize])"""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    inv = 1.0 / filterscale
    one = 1 << _PRECISION_BITS
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * inv) for x in range(xmax)]
        tot = 0.0
        for v in w:
            tot += v
        for x in range(xmax):
            k = w[x] / tot if tot != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * one) if k < 0 else int(0.5 + k * one)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _tables:
        ks, b, k = _coeffs(in_size, out_size)
        _tables[key] = (ks, torch.from_numpy(b).to(device), torch.from_numpy(k).to(device))
    return _tables[key]


def _as_u8_hwc(img, width=None, height=None):
    if torch.is_tensor(img):
        a = img
        if a.dtype != torch.uint8 or a.dim() != 3 or a.shape[2] != 3:
            raise ValueError("tensor images must be uint8 [H, W, 3] (RGB)")
        return a
    if hasattr(img, "convert"):                 # PIL image: do_convert_rgb
        # diffusers' VaeImageProcessor resizes BEFORE convert("RGB"): RGB and L sources resample identically either way (the GPU path
        # below, bit-exact); palette ("P": nearest-neighbour resize) and alpha ("RGBA" / "LA": premultiplied resampling) sources do
        # not, so for those the reference's own order runs on the host -- Pillow's resize in the SOURCE mode, then the conversion --
        # and the GPU pass only does /255, layout and the repeats (they used to be refused)
        if getattr(img, "mode", "RGB") not in ("RGB", "L"):
            if width is not None and img.size != (width, height):
                from PIL import Image
                img = img.resize((width, height), resample=getattr(Image, "Resampling", Image).LANCZOS)
        img = np.asarray(img.convert("RGB"))
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("images must be PIL images or uint8 [H, W, 3] arrays")
    return torch.from_numpy(np.ascontiguousarray(a))


@torch.no_grad()
def prepare_images(images, width, height, batch_size, num_images_per_prompt, device, dtype,
                   do_classifier_free_guidance=False, guess_mode=False):
    """-> [1 (2 with CFG and not guess_mode), len(images) * batch_size * num_images_per_prompt, 3, height, width] on `device`"""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("prepare_images (libctrlhip) runs on the GPU only; there is no CPU fallback")
    # diffusers' get_default_height_width rounds both down to a multiple of the VAE scale factor (8), as the reference does
    # (model/ctrl_helper.py:268-296 -> VaeImageProcessor.preprocess); every size the reference's scripts use (512, 1024) already is one
    width, height = int(width) - int(width) % 8, int(height) - int(height) % 8
    if width <= 0 or height <= 0:
        raise ValueError("prepare_images: width and height must be at least 8")
    frames = [_as_u8_hwc(i, width, height) for i in images]
    if not frames or any(f.shape != frames[0].shape for f in frames):
        raise ValueError("prepare_images needs at least one image and all images of one size")
    src = torch.stack([f.to(device) for f in frames]).contiguous()        # [F][Hin][Win][3] uint8
    F_, Hin, Win, _ = src.shape
    rep = int(batch_size) * int(num_images_per_prompt)
    cfg = 2 if (do_classifier_free_guidance and not guess_mode) else 1
    out = torch.empty(cfg, rep * F_, 3, height, width, dtype=dtype, device=device)
    hb = hk = vb = vk = tmp = None
    hks = vks = 0
    if Win != width:
        hks, hb, hk = _device_tables(Win, width, device)
        tmp = torch.empty(F_, Hin, width, 3, dtype=torch.uint8, device=device)
    if Hin != height:
        vks, vb, vk = _device_tables(Hin, height, device)
    with torch.cuda.device(device):
        L.check(L.lib().ctrl_prepare_images(L.ptr(src), F_, Hin, Win, L.ptr(hb), L.ptr(hk), hks, L.ptr(vb), L.ptr(vk), vks,
                                            L.ptr(tmp), L.ptr(out), L.dtype_code(dtype), width, height, rep, cfg, L.cur_stream()))
    return out
