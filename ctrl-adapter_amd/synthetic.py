"""Seeded, NAME-keyed synthetic weights / inputs (no checkpoints or datasets are reachable offline): used by bench.py,
the golden generator and the tests (oracle/init.py re-exports this module; nothing here imports oracle/).  Every parameter is drawn from a generator seeded by crc32(name) ^ seed, so two module
trees with the same state-dict keys (the reference's own classes and the oracle restatement) receive
identical weights regardless of construction order -- and a key mismatch shows up as a value mismatch.

Scales: >=2-D weights ~ N(0, 1/fan_in) (unit gain, so every branch -- attention logits, zero-initialised
ControlNet convs included -- carries signal and parity is not vacuous); biases ~ N(0, 0.1^2); norm weights
1 + N(0, 0.1^2); AlphaBlender mix_factor = 0.5; router wg ~ N(0, 1).  All values are rounded to fp16-representable
numbers so the fp16 HIP path and the fp32 oracle consume bit-identical weights.
"""
import zlib

import torch


@torch.no_grad()
def seeded_init(module, seed=1234, fp16_round=True, gain=1.0, dist="normal", gamma_outliers=0.0, gamma_outlier_scale=8.0):
    """gain / dist / gamma_outliers: the weight-distribution sweep of tests/test_gpu_e2e.py (a trained checkpoint has heavier tails and
    outlier channels than unit-gain Gaussians): `gain` multiplies the standard deviation of every >= 2-D weight; dist="student4" draws
    them from a Student-t with 4 degrees of freedom scaled to the same variance; `gamma_outliers` = fraction of the channels of every
    normalisation scale (GroupNorm and LayerNorm weights alike) that is multiplied by gamma_outlier_scale."""
    for name, p in module.named_parameters():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "mix_factor":
            v = torch.full(p.shape, 0.5)
        elif ".wg." in name or name.startswith("wg."):
            v = torch.randn(p.shape, generator=g)
        elif p.dim() >= 2:
            fan_in = p[0].numel()
            if dist == "student4":
                # t_4 = z / sqrt(chi2_4 / 4), variance 4 / (4 - 2) = 2: divided by sqrt(2) to unit variance
                z = torch.randn(p.shape, generator=g)
                chi = torch.randn((4,) + tuple(p.shape), generator=g).pow(2).sum(0)
                v = z / (chi / 4.0).sqrt() / 2.0 ** 0.5
                v = v / fan_in ** 0.5
            else:
                v = torch.randn(p.shape, generator=g) / fan_in ** 0.5      # (exactly this expression: the committed goldens were drawn with it)
            if gain != 1.0:
                v = v * gain
        elif leaf == "weight":            # 1-D weight = GroupNorm / LayerNorm scale
            v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            if gamma_outliers > 0.0:
                pick = torch.rand(p.shape, generator=g) < gamma_outliers
                v = torch.where(pick, v * gamma_outlier_scale, v)
        else:
            v = 0.1 * torch.randn(p.shape, generator=g)
        if fp16_round:
            v = v.half().float()
        p.copy_(v.to(p.dtype))
    return module


def seeded_tensor(shape, seed, kind="normal", fp16_round=True):
    g = torch.Generator().manual_seed(seed)
    v = torch.rand(shape, generator=g) if kind == "uniform" else torch.randn(shape, generator=g)
    return v.half().float() if fp16_round else v
