"""Instruction mix of the MFMA-carrying basic blocks of a kernel, from the device ISA hipcc emits for gfx950 (no GPU needed):

  python tools/isa_mix.py ctrl-adapter_amd/csrc/attention_d64.hip 'flash_attn_d64_kernel<1, 8, 8, 3>' [out.txt]
  python tools/isa_mix.py ctrl-adapter_amd/csrc/igemm.hip 'igemm_kernel<256, 256, 32, 2, 4, 4, 0, true>'

For every basic block that holds matrix instructions: how many MFMA, VALU (transcendental ones apart), LDS, VMEM / LDS-DMA, scalar
waits, barriers and spill instructions it issues -- the "VALU per MFMA" figures of DESIGN.md section 8 and
profiles/r03_attention_ceiling.md, and the check that a loop body carries no scratch traffic.  Register / scratch totals of the
kernel come from the code-object metadata of the same compilation."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_traffic import symbol_of  # noqa: E402

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-w",
         "-I" + os.path.join(ROOT, "include"), "-x", "hip", "--cuda-device-only", "-S"]
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def device_asm(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [src, "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("hipcc failed:\n" + r.stderr[-2000:])
        return open(out).read()


def classify(op):
    if "mfma" in op:
        return "mfma"
    if op.startswith("scratch_") or op.startswith(("v_readlane", "v_writelane")):
        return "spill"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load_lds", "buffer_load")) and "lds" in op:
        return "lds_dma"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op == "s_barrier":
        return "s_barrier"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(asm):
    """name -> body text, and name -> metadata dict"""
    bodies = {}
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n", asm):
        end = asm.find(".Lfunc_end", m.end())
        bodies[m.group(1)] = asm[m.end():end]
    meta = {}
    md = asm[asm.find("amdhsa.kernels"):]
    for e in md.split("- .agpr_count")[1:]:
        n = re.search(r"\.name:\s+(\S+)", e).group(1)
        meta[n] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1)) for k in
                   ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    return bodies, meta


def report(src, want, out=None):
    asm = device_asm(src)
    bodies, meta = kernels(asm)
    lines = []
    hit = False
    for name, body in bodies.items():
        sym = symbol_of(name)
        if want and re.sub(r"\s+", "", want) != re.sub(r"\s+", "", sym):
            continue
        hit = True
        md = meta.get(name, {})
        lines.append("%s\n  registers %s  SGPR %s  scratch %s B" % (sym, md.get("vgpr_count"), md.get("sgpr_count"), md.get("private_segment_fixed_size")))
        blocks = re.split(r"\n(\.LBB\d+_\d+):", "\n.LBBentry_0:" + body)
        for j in range(1, len(blocks), 2):
            ops = [ln.strip().split()[0] for ln in blocks[j + 1].splitlines() if ln.strip() and not ln.strip().startswith((";", "."))]
            c = collections.Counter(classify(o) for o in ops)
            if not c["mfma"]:
                continue
            valu = c["valu"] + c["valu_trans"]
            lines.append("  %-12s %4d instr: MFMA %3d  VALU %3d (transcendental %2d)  LDS %2d  LDS-DMA %d  VMEM %d  s_waitcnt %2d  s_barrier %d  "
                         "s_nop %2d  spill %d   VALU per MFMA %.1f" % (blocks[j], len(ops), c["mfma"], valu, c["valu_trans"], c["lds"], c["lds_dma"],
                                                                      c["vmem"], c["s_waitcnt"], c["s_barrier"], c["s_nop"], c["spill"], valu / c["mfma"]))
    if not hit:
        raise SystemExit("no kernel spelled %r in %s; available:\n  %s" % (want, src, "\n  ".join(sorted(set(symbol_of(n) for n in bodies)))))
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if out:
        with open(out, "w") as fh:
            fh.write("# python tools/isa_mix.py %s %r\n" % (os.path.relpath(src, ROOT), want))
            fh.write(text)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    report(os.path.abspath(sys.argv[1]), sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None)
