#!/usr/bin/env python3
"""Register / spill table of every igemm_kernel instantiation (no GPU needed: hipcc cross-compiles gfx950).

    python tools/igemm_resources.py > profiles/rNN_igemm_kernel_resources.txt

Per instantiation: what `hipcc -Rpass-analysis=kernel-resource-usage` reports, and the number of spill instructions
(v_readlane / v_writelane / scratch_*) inside the tightest loop of the device ISA that contains the MFMAs -- the k-loop.  A
register-tight kernel may spill in its epilogue without harm; a spill in the k-loop is a performance bug."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ctrl-adapter_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-I" + os.path.join(ROOT, "include")]


def pretty(name):
    m = re.search(r"igemm8_kernelILi(\d+)ELi(\d+)E", name)
    if m:
        return "igemm8_kernel<%s, %s>" % m.groups()
    m = re.search(r"igemm_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])E", name)
    if not m:
        return re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:40]
    g = m.groups()
    return "igemm_kernel<%s, %s, %s, %s, %s, %s, %s, %s>" % (g[:7] + ("true" if g[7] == "1" else "false",))


def analyse(src):
    """ONE device-only compilation: the resource remarks (stderr) and the ISA (-S); per kernel the remark fields and the
    tightest loop that holds MFMAs as (instructions, MFMAs, spill ops)"""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "x.s")
        r = subprocess.run([HIPCC] + FLAGS + ["--cuda-device-only", "-S", src, "-o", out, "-Rpass-analysis=kernel-resource-usage"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=CSRC)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout.decode()[-3000:])
        txt = open(out).read()
    return parse(r.stdout.decode(), txt)


def parse(remarks, txt):
    """remarks: hipcc's -Rpass-analysis=kernel-resource-usage output; txt: the device ISA of the same compilation"""
    res, cur = {}, None
    keys = (("VGPRs", "vgpr"), ("TotalSGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"), ("SGPRs Spill", "sgpr_spill"),
            ("VGPRs Spill", "vgpr_spill"), ("Occupancy [waves/SIMD]", "occ"))
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
        for k, kk in keys:
            mm = re.search(re.escape(k) + r": (\d+)", line)
            if mm and cur:
                res[cur][kk] = int(mm.group(1))
    loops = {}
    for fn in re.split(r"\n(?=_ZN12_GLOBAL__N_1\d+\w+:)", txt):
        m = re.match(r"(_ZN12_GLOBAL__N_1\d+\w+):", fn)
        if not m:
            continue
        lines, labels, best = fn.split("\n"), {}, None
        for i, line in enumerate(lines):
            mm = re.match(r"^(\.LBB\d+_\d+):", line)
            if mm:
                labels[mm.group(1)] = i
        for i, line in enumerate(lines):
            mm = re.search(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", line)
            if mm and mm.group(2) in labels and labels[mm.group(2)] < i:          # a backward branch closes a loop
                body = lines[labels[mm.group(2)]:i]
                nm = sum("v_mfma" in x for x in body)
                if nm and (best is None or len(body) < best[0]):
                    best = (len(body), nm, sum(("v_readlane" in x or "v_writelane" in x or "scratch_" in x) for x in body))
        loops[m.group(1)] = best
    for n in res:
        res[n]["loop"] = loops.get(n)
    return res


def violations(res):
    """what tests/test_kernel_resources.py fails on: (a) a spill instruction inside the tightest MFMA loop (a reload there waits on
    vmcnt and drains the LDS-DMA queue); (b) scratch without register spills = an array or the kernel-argument descriptor living in
    private memory (accumulators indexed by a register, `select between loaded argument fields` turned into a load from a selected
    address -- both happened in round 4, both cost every access a vmcnt wait)"""
    bad = []
    for n, v in sorted(res.items()):
        if "igemm" not in n:
            continue
        lp = v.get("loop")
        if lp and lp[2] > 0:
            bad.append("%s: %d spill instruction(s) inside its MFMA loop" % (pretty(n), lp[2]))
        if v.get("scratch", 0) > 4 * v.get("vgpr_spill", 0) + 16:
            bad.append("%s: %d B of scratch for %d spilled registers: something lives in private memory" % (pretty(n), v.get("scratch", 0), v.get("vgpr_spill", 0)))
    return bad


def main():
    src = os.path.join(CSRC, "igemm.hip")
    res = analyse(src)
    print("# igemm.hip: hipcc -Rpass-analysis=kernel-resource-usage per instantiation, and the number of spill instructions")
    print("# (v_readlane / v_writelane / scratch_*) inside the tightest loop that contains the MFMAs (device ISA).  tools/igemm_resources.py")
    print("%-62s %5s %5s %4s %8s %11s %11s  %s" % ("kernel", "VGPR", "SGPR", "occ", "scratch", "SGPR spill", "VGPR spill", "tightest MFMA loop: instr / MFMAs / spill ops"))
    for n, v in sorted(res.items(), key=lambda kv: pretty(kv[0])):
        lp = v.get("loop")
        print("%-62s %5d %5d %4d %8d %11d %11d  %s" % (pretty(n), v.get("vgpr", 0), v.get("sgpr", 0), v.get("occ", 0), v.get("scratch", 0),
                                                         v.get("sgpr_spill", 0), v.get("vgpr_spill", 0), "%d / %d / %d" % lp if lp else "-"))
    bad = violations(res)
    for b in bad:
        print("# VIOLATION: " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
