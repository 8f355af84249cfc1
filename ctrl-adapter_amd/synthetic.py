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
def seeded_init(module, seed=1234, fp16_round=True):
    for name, p in module.named_parameters():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "mix_factor":
            v = torch.full(p.shape, 0.5)
        elif ".wg." in name or name.startswith("wg."):
            v = torch.randn(p.shape, generator=g)
        elif p.dim() >= 2:
            fan_in = p[0].numel()
            v = torch.randn(p.shape, generator=g) / fan_in ** 0.5
        elif leaf == "weight":            # 1-D weight = GroupNorm / LayerNorm scale
            v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
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
