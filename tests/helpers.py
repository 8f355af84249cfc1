import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLD):
    if p not in sys.path:
        sys.path.insert(0, p)


def load_golden(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def check_digest(t, d, tol, what=""):
    """compare a tensor with a stored digest (tests/golden/make_golden.py): strided samples + sums; returns rel-inf"""
    t = t.detach().float().cpu().contiguous()
    assert list(t.shape) == d["shape"], "%s: shape %s != golden %s" % (what, list(t.shape), d["shape"])
    flat = t.reshape(-1)
    s = flat[::d["step"]][:4096]
    ref = d["samples"]
    denom = max(ref.abs().max().item(), 1e-12)
    err = ((s - ref).abs().max().item()) / denom
    assert err <= tol, "%s: rel_inf %.3e > %.1e" % (what, err, tol)
    # whole-tensor check through the abs-sum (catches errors outside the sampled positions)
    rel_sum = abs(float(flat.double().abs().sum()) - d["abssum"]) / max(d["abssum"], 1e-12)
    assert rel_sum <= max(tol, 1e-6) * 2, "%s: abs-sum differs by %.3e" % (what, rel_sum)
    if "full" in d:
        # the whole tensor is stored: element-by-element rel-inf (supersedes the strided samples)
        full = d["full"].float()
        err = max(err, ((t - full).abs().max().item()) / max(full.abs().max().item(), 1e-12))
        assert err <= tol, "%s: rel_inf over the full tensor %.3e > %.1e" % (what, err, tol)
    return err
