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


def test_ffn_kernel_default_build_has_no_spills_and_one_workgroup_of_registers():
    """csrc/ffn.hip (round 6): the default instantiation of the fused feed-forward kernel holds S (64) + O (128) accumulator registers and
    its operand fragments in the 256 architected registers of a one-wave-per-SIMD workgroup: no spill anywhere (hipcc's habit with this
    kernel, DESIGN.md section 3: loop-invariant addresses hoisted to kernel entry and spilled around the chunk loop) and nothing in scratch.
    Cached per source revision like the implicit-GEMM audit."""
    import hashlib
    tool = _tool()
    csrc = os.path.join(ROOT, "ctrl-adapter_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".h") or f.endswith(".inc") or f == "ffn.hip":
            h.update(open(os.path.join(csrc, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "ctrl_hip.h"), "rb").read())
    h.update(" ".join(tool.FLAGS).encode())
    digest = h.hexdigest()
    cache = os.path.join(ROOT, "ctrl-adapter_amd", "build", "ffn_resources.json")
    res = None
    if os.path.exists(cache):
        c = json.load(open(cache))
        if c.get("digest") == digest:
            res = c["res"]
    if res is None:
        res = tool.analyse(os.path.join(csrc, "ffn.hip"))
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        json.dump({"digest": digest, "res": res}, open(cache, "w"))
    main = [n for n in res if "ffn512_kernel" in n and "ILi0E" in n]          # MODE 0 = FF_PLAIN, the product build
    assert len(main) == 1, sorted(res)
    r = res[main[0]]
    assert r["vgpr_spill"] == 0 and r["scratch"] == 0, r                         # (SGPR spills go to VGPR lanes: the epilogue's descriptor fields)
    assert r["vgpr"] <= 256 and r["occ"] >= 1, r
    assert r.get("loop"), r                                                     # the chunk loop was found and holds MFMAs
    ins, mfma, spills = r["loop"][:3]
    assert mfma == 384 and spills == 0, r["loop"]                               # 256 (S = X W1c^T) + 128 (O += P W2c) MFMAs per chunk; nothing spilled inside
