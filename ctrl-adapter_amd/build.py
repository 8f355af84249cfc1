"""Builds libctrlhip.so (hand-written gfx950 HIP kernels + the C-ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the
resulting .so ships to the GPU box with the source tree (git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libctrlhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Wno-unused-result"]


# per-source additions to FLAGS.  The attention loops: hipcc's SLP vectoriser turns adjacent f32 adds / multiplies of the online softmax (row sums,
# the rescale) into v_pk_add_f32 / v_pk_mul_f32, which MI355X_MICROARCH.md prices as an anti-lever beside MFMAs.  Same-box A/B (round 6,
# profiles/r06_ab_slp.txt): the 128^2 self-attention launch 8.34 -> 8.20 ms in the step (-1.6 %), 2.81 -> 2.78 ms alone; the fused feed-forward
# (whose erf polynomial is packed the same way) does not move (0.901 vs 0.903 ms) and keeps the default flags.  CTRL_BUILD_SLP=1 builds the
# attention files with the vectoriser on (the rounds 2-5 build) for A/B runs.
EXTRA_FLAGS = {}
if os.environ.get("CTRL_BUILD_SLP", "0") != "1":
    for _f in ("attention_d64.hip", "attention.hip"):
        EXTRA_FLAGS[_f] = ["-fno-slp-vectorize"]


if os.environ.get("CTRL_BUILD_FMAXF", "0") == "1":          # A/B: fmaxf (maxnum + canonicalising v_max) instead of v_maximum3_f32 in the attention loops
    for _f in ("attention_d64.hip", "attention.hip"):
        EXTRA_FLAGS[_f] = EXTRA_FLAGS.get(_f, []) + ["-DCTRL_ATTN_FMAXF"]


if os.environ.get("CTRL_BUILD_ATTN_SCHED"):          # A/B: LLVM's AMDGPU scheduling strategy for the attention files (max-ilp | max-memory-clause | iterative-ilp)
    for _f in ("attention_d64.hip", "attention.hip"):
        EXTRA_FLAGS[_f] = EXTRA_FLAGS.get(_f, []) + ["-mllvm", "-amdgpu-sched-strategy=" + os.environ["CTRL_BUILD_ATTN_SCHED"]]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest(path):
    h = hashlib.sha1()
    for dep in [path] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
            [os.path.join(HERE, "..", "include", "ctrl_hip.h")]:
        with open(dep, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(os.path.basename(path), [])).encode())
    return h.hexdigest()


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + ".o")
    stamp = obj + ".sha1"
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-x", "hip", "-c", path, "-o", obj]
    audit = src == "igemm.hip" and os.environ.get("CTRL_BUILD_AUDIT", "1") != "0"      # (=0: faster edit-compile cycles)
    if audit:
        # the register / spill audit of the implicit-GEMM kernels (tools/igemm_resources.py, tests/test_kernel_resources.py) rides on
        # this compilation: its remarks and the device ISA it leaves behind, instead of a second three-minute compile
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if audit:
        _write_audit(r.stderr)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return obj, True


def audit_digest():
    """what the cached audit is valid for: igemm.hip, the headers, the flags (tests/test_kernel_resources.py recomputes it)"""
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h") or f == "igemm.hip":
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    with open(os.path.join(HERE, "..", "include", "ctrl_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _write_audit(remarks):
    import glob
    import importlib.util
    import json
    try:
        spec = importlib.util.spec_from_file_location("igemm_resources", os.path.join(HERE, "..", "tools", "igemm_resources.py"))
        tool = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(tool)
        asm = glob.glob(os.path.join(OBJ, "igemm*gfx950*.s"))
        if asm:
            res = tool.parse(remarks, open(asm[0]).read())
            with open(os.path.join(OBJ, "igemm_resources.json"), "w") as fh:
                json.dump({"digest": audit_digest(), "res": res}, fh)
    except Exception as e:       # the audit is a by-product: never fail the build over it
        print("igemm audit not written: %s" % e)
    for f in glob.glob(os.path.join(OBJ, "igemm*.bc")) + glob.glob(os.path.join(OBJ, "igemm*.hipi")) + glob.glob(os.path.join(OBJ, "igemm*.out*")) + \
            glob.glob(os.path.join(OBJ, "igemm*.hipfb")) + glob.glob(os.path.join(OBJ, "igemm*-host-*")) + glob.glob(os.path.join(OBJ, "igemm*gfx950*.o")):
        try:
            os.remove(f)
        except OSError:
            pass


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        res = list(ex.map(_compile, sources()))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("libctrlhip: %d sources, %s -> %s" % (len(objs), "rebuilt" if changed else "up to date", LIB))
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
