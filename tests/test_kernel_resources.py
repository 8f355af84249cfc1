"""No GPU needed: the implicit-GEMM kernels must not spill inside their MFMA loops, and nothing of theirs may live in scratch
(tools/igemm_resources.py; round-3 review item 8).  hipcc cross-compiles gfx950 here; the analysis of one source revision is
cached next to the build's object files, so the ~1 min compilation is paid once per change of igemm.hip."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("igemm_resources", os.path.join(ROOT, "tools", "igemm_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_igemm_kernels_have_no_spills_in_mfma_loops_and_nothing_in_scratch():
    tool = _tool()
    csrc = os.path.join(ROOT, "ctrl-adapter_amd", "csrc")
    spec = importlib.util.spec_from_file_location("_ctrl_build", os.path.join(ROOT, "ctrl-adapter_amd", "build.py"))
    bld = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bld)
    digest = bld.audit_digest()
    cache = os.path.join(ROOT, "ctrl-adapter_amd", "build", "igemm_resources.json")
    res = None
    if os.path.exists(cache):              # written by build.py as a by-product of compiling igemm.hip
        c = json.load(open(cache))
        if c.get("digest") == digest:
            res = c["res"]
    if res is None:
        res = tool.analyse(os.path.join(csrc, "igemm.hip"))
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        json.dump({"digest": digest, "res": res}, open(cache, "w"))
    kernels = [n for n in res if "igemm" in n and "splitk" not in n]
    assert len(kernels) >= 20, kernels
    assert any("igemm8" in n for n in kernels)
    # every MFMA kernel reports its k-loop
    for n in kernels:
        assert res[n].get("loop"), n
    bad = tool.violations(res)
    assert not bad, "\n".join(bad)
