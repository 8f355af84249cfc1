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
    m = re.search(r"igemm_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])E", name)
    if not m:
        return re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:40]
    g = m.groups()
    tf = lambda v: "true" if v == "1" else "false"
    return "igemm_kernel<%s, %s, %s, %s, %s, %s, %s, %s, %s>" % (g[:7] + (tf(g[7]), tf(g[8])))


def resource_remarks(src):
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([HIPCC] + FLAGS + ["-c", src, "-o", os.path.join(d, "x.o"), "-Rpass-analysis=kernel-resource-usage"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=CSRC)
    res, cur = {}, None
    keys = (("VGPRs", "vgpr"), ("TotalSGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"), ("SGPRs Spill", "sgpr_spill"),
            ("VGPRs Spill", "vgpr_spill"), ("Occupancy [waves/SIMD]", "occ"))
    for line in r.stdout.decode().splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
        for k, kk in keys:
            mm = re.search(re.escape(k) + r": (\d+)", line)
            if mm and cur:
                res[cur][kk] = int(mm.group(1))
    return res


def mfma_loops(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "x.s")
        subprocess.run([HIPCC] + FLAGS + ["--cuda-device-only", "-S", src, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=CSRC, check=True)
        txt = open(out).read()
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
    return loops


def main():
    src = os.path.join(CSRC, "igemm.hip")
    res, loops = resource_remarks(src), mfma_loops(src)
    print("# igemm.hip: hipcc -Rpass-analysis=kernel-resource-usage per instantiation, and the number of spill instructions")
    print("# (v_readlane / v_writelane / scratch_*) inside the tightest loop that contains the MFMAs (device ISA).  tools/igemm_resources.py")
    print("# (scratch > 0 only in the BK = 64 256x128 tile of channel counts that are multiples of 64 but not 128: epilogue spills, outside the MFMA loop)")
    print("%-62s %5s %5s %4s %8s %11s %11s  %s" % ("kernel", "VGPR", "SGPR", "occ", "scratch", "SGPR spill", "VGPR spill", "tightest MFMA loop: instr / MFMAs / spill ops"))
    for n, v in sorted(res.items(), key=lambda kv: pretty(kv[0])):
        lp = loops.get(n)
        print("%-62s %5d %5d %4d %8d %11d %11d  %s" % (pretty(n), v.get("vgpr", 0), v.get("sgpr", 0), v.get("occ", 0), v.get("scratch", 0),
                                                         v.get("sgpr_spill", 0), v.get("vgpr_spill", 0), "%d / %d / %d" % lp if lp else "-"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
