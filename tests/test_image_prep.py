"""SURVEY.md 8f row 3 (host-side input prep): the oracle restatement of the conditioning-image preparation is pinned bit for
bit against Pillow itself -- the third-party library whose 8-bit Lanczos resampling is the only arithmetic of
model/ctrl_helper.py:268-296 -- on CPU; the HIP implementation is compared with the oracle under -m gpu."""
import numpy as np
import pytest
import torch

from oracle import image_prep as O

PIL = pytest.importorskip("PIL.Image")


def _rand_img(h, w, seed):
    g = np.random.default_rng(seed)
    base = g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    base[: h // 3, : w // 2] = 255          # flat and saturated regions: overshoot / clipping of the negative Lanczos lobes
    base[h // 2:, w // 2:] = 0
    return base


@pytest.mark.parametrize("h,w,oh,ow", [(64, 96, 64, 64), (100, 75, 64, 48), (40, 40, 96, 128), (123, 77, 50, 31), (48, 48, 48, 48)])
def test_oracle_resize_is_bit_exact_with_pillow(h, w, oh, ow):
    img = _rand_img(h, w, seed=h * 1000 + w)
    ref = np.asarray(PIL.fromarray(img, "RGB").resize((ow, oh), resample=PIL.Resampling.LANCZOS))
    got = O.pil_resize_lanczos(img, ow, oh)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_oracle_prepare_images_matches_the_reference_recipe():
    """the recipe of model/ctrl_helper.py:280-294 written out with PIL + torch, against the oracle"""
    imgs = [_rand_img(80, 60, seed=s) for s in range(3)]
    W = H = 64
    pre = []
    for im in imgs:
        p = PIL.fromarray(im, "RGB").convert("RGB").resize((W, H), resample=PIL.Resampling.LANCZOS)
        pre.append(torch.from_numpy(np.array(p).astype(np.float32) / 255.0).permute(2, 0, 1)[None].to(torch.float32))
    x = torch.cat(pre, dim=0).repeat(2, 1, 1, 1).unsqueeze(0).to(torch.float16).repeat(2, 1, 1, 1, 1)
    got = O.prepare_images(imgs, W, H, batch_size=2, num_images_per_prompt=1, dtype=torch.float16, do_classifier_free_guidance=True)
    assert got.shape == (2, 6, 3, H, W) and torch.equal(got, x)
    assert O.prepare_images(imgs, W, H, 1, 1, do_classifier_free_guidance=True, guess_mode=True).shape == (1, 3, 3, H, W)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,oh,ow,dt", [(512, 512, 512, 512, torch.float16), (300, 400, 512, 512, torch.float16),
                                          (720, 1280, 512, 512, torch.float32), (64, 48, 96, 40, torch.bfloat16)])
def test_hip_prepare_images_is_bit_exact(gpu, h, w, oh, ow, dt):
    import ctrl_adapter_amd as P
    imgs = [_rand_img(h, w, seed=7 * s + h) for s in range(4)]
    ref = O.prepare_images(imgs, ow, oh, batch_size=2, num_images_per_prompt=1, dtype=dt, do_classifier_free_guidance=True)
    got = P.prepare_images([PIL.fromarray(i, "RGB") for i in imgs], ow, oh, 2, 1, gpu, dt, do_classifier_free_guidance=True)
    assert got.shape == ref.shape and got.dtype == dt and torch.equal(got.cpu(), ref)
    flat = P.prepare_images([torch.from_numpy(i) for i in imgs], ow, oh, 1, 1, gpu, dt)          # uint8 tensors in, no CFG
    assert torch.equal(flat.cpu(), O.prepare_images(imgs, ow, oh, 1, 1, dtype=dt))
    print("PARITY prepare_images %dx%d -> %dx%d %s: bit-exact" % (h, w, oh, ow, dt))
    # sizes are rounded DOWN to a multiple of 8, as diffusers' get_default_height_width does (100 x 96 -> 96 x 96)
    r8 = P.prepare_images([torch.from_numpy(imgs[0])], 100, 96, 1, 1, gpu, dt)
    assert r8.shape[-2:] == (96, 96) and torch.equal(r8.cpu(), O.prepare_images(imgs[:1], 96, 96, 1, 1, dtype=dt))
    # palette / alpha sources: the reference resizes in the SOURCE mode, then converts (VaeImageProcessor.preprocess) -- written
    # out with Pillow here
    for mode in ("RGBA", "P", "LA"):
        src = PIL.fromarray(imgs[0], "RGB").convert(mode)
        want = np.array(src.resize((ow, oh), resample=PIL.Resampling.LANCZOS).convert("RGB")).astype(np.float32) / 255.0
        want = torch.from_numpy(want).permute(2, 0, 1)[None, None].to(dt)
        got_m = P.prepare_images([src], ow, oh, 1, 1, gpu, dt)
        assert got_m.shape == want.shape and torch.equal(got_m.cpu(), want), mode
