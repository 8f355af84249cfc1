"""Which `diffusers` the reference's files run on when the goldens are made / the live check runs.

The reference imports its transformer / ResNet blocks from diffusers (model/adapter_spatial_temporal.py:96-134,
controlnet/controlnet.py:371-424, model/resnet_block_2d.py:28).  Its requirements_inference.txt names `diffusers` WITHOUT a version; 0.27.2
(PINNED below) is inferred from the "copied from .../diffusers/blob/v0.27.2/..." headers of the reference's own files.  That package is not
installable in the build container (no network), so the goldens committed here were made over oracle/_shim, a minimal
`diffusers` whose blocks ARE oracle/blocks.py: they pin the reference's own glue and are circular for the diffusers arithmetic.
This module closes that: when a REAL diffusers is importable it is used instead of the shim, and every golden file carries
which one made it.

    select(require_real=False) -> provenance string, with sys.path prepared so that `import diffusers` resolves accordingly
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHIM = os.path.join(ROOT, "oracle", "_shim")
PINNED = "0.27.2"


def _real_diffusers():
    """the installed diffusers, imported with the shim OFF the path; None when there is none"""
    saved = list(sys.path)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != SHIM]
    for k in [k for k in sys.modules if k == "diffusers" or k.startswith("diffusers.")]:
        del sys.modules[k]
    try:
        mod = importlib.import_module("diffusers")
        f = os.path.abspath(getattr(mod, "__file__", "") or "")
        if f.startswith(SHIM) or not hasattr(mod, "__version__"):
            raise ImportError("that is the shim")
        return mod
    except Exception:
        for k in [k for k in sys.modules if k == "diffusers" or k.startswith("diffusers.")]:
            del sys.modules[k]
        sys.path[:] = saved
        return None


def select(require_real=False):
    mod = _real_diffusers()
    if mod is not None:
        return "diffusers==%s" % mod.__version__
    if require_real:
        raise SystemExit("--require-real-diffusers: no real `diffusers` is importable here (pip install diffusers==%s); only the shim "
                         "oracle/_shim is available" % PINNED)
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    return "oracle-shim (oracle/blocks.py restating diffusers v%s)" % PINNED


def provenance(blocks):
    import torch
    return {"blocks": blocks, "torch": torch.__version__, "generator": "tests/golden/make_golden.py", "pinned_diffusers": PINNED}
