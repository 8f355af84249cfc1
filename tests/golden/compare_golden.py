"""Compares two directories of golden files (tests/golden/make_golden.py): the committed ones against a regeneration, e.g. one made
over a REAL diffusers (tests/golden/README.md).  Tensors are compared by rel-inf (max|a-b| / max|b|), everything else exactly.

    python tests/golden/compare_golden.py DIR_A DIR_B [tolerance, default 0 = bit-identical]
"""
import os
import sys

import torch

SKIP = ("provenance", "__provenance__")


def walk(a, b, path, tol, worst):
    if isinstance(a, dict):
        keys = [k for k in a if k not in SKIP]
        assert sorted(map(str, keys)) == sorted(str(k) for k in b if k not in SKIP), "%s: keys differ" % path
        for k in keys:
            walk(a[k], b[k], "%s/%s" % (path, k), tol, worst)
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), "%s: length %d != %d" % (path, len(a), len(b))
        for i, (x, y) in enumerate(zip(a, b)):
            walk(x, y, "%s/%d" % (path, i), tol, worst)
    elif torch.is_tensor(a):
        assert a.shape == b.shape, "%s: shape" % path
        den = max(b.double().abs().max().item(), 1e-30) if b.numel() else 1.0
        e = ((a.double() - b.double()).abs().max().item() / den) if a.numel() else 0.0
        worst[0] = max(worst[0], e)
        assert e <= tol, "%s: rel-inf %.3e > %.1e" % (path, e, tol)
    elif isinstance(a, float):
        assert abs(a - b) <= max(tol, 1e-12) * max(abs(b), 1.0) * 4, "%s: %r != %r" % (path, a, b)
    else:
        assert a == b, "%s: %r != %r" % (path, a, b)


def main():
    da, db = sys.argv[1], sys.argv[2]
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    files = sorted(f for f in os.listdir(da) if f.endswith(".pt"))
    assert files, "no .pt files in %s" % da
    for f in files:
        a = torch.load(os.path.join(da, f), weights_only=False)
        b = torch.load(os.path.join(db, f), weights_only=False)
        worst = [0.0]
        walk(a, b, f, tol, worst)
        pa = a.get("provenance") or a.get("__provenance__") or {}
        pb = b.get("provenance") or b.get("__provenance__") or {}
        print("%-32s worst rel-inf %.2e   [%s]  vs  [%s]" % (f, worst[0], pa.get("blocks", "?"), pb.get("blocks", "?")))
    print("GOLDENS AGREE (tolerance %.1e)" % tol)


if __name__ == "__main__":
    main()
