"""Generates (and checks) the chunk-loop body of the fused feed-forward kernel, csrc/ffn_body_<name>.inc.

A chunk is 24 phases (G1 k-tile t = phases 2t, 2t+1; G2 k-step kap = phases 16 + 2 kap, + 1) over 32 PIECES of 16 KB that go round an
8-slot LDS ring (slot = piece % 8):  piece 3t + {0, 1, 2} = X / W1 rows 0-127 / W1 rows 128-255 of G1 k-tile t (first read in phase 2t,
last read in phase 2t + 1);  piece 24 + 4f + q = W2 rows 128q.. of G2 k-tile f (phases 16 + 4f .. 16 + 4f + 3).  Pieces 32 + i are pieces
i of the NEXT chunk.  A schedule says in which phase every piece 8 .. 39 is handed to the LDS-DMA (pieces 0 .. 7 of the first chunk are
staged by the kernel's prologue); this script

  * checks the two hazards of the ring against it --
      WAR: piece j goes where piece j - 8 was: it is issued at least ONE phase after the last phase that read piece j - 8 (every wave
           retires its fragment reads before the first barrier of a phase, the DMA is issued in the load segment of a later phase);
      RAW: piece j is waited for (counted s_waitcnt vmcnt) in the phase BEFORE its first read -- the other wave row's barrier lies in
           between -- so it is issued in that phase at the latest;
  * derives the vmcnt immediates: at the wait of phase w every piece first read in phase <= w + 1 has landed, i.e. at most
    2 x (pieces issued so far that are first read later) loads are outstanding (2 DMA instructions per piece and wave, in-order counter);
  * writes the loop body (macros of ffn.hip).

    python tools/gen_ffn_schedule.py            # rewrites the .inc files
    python tools/gen_ffn_schedule.py --check    # exit 1 if a committed .inc differs from what the schedule gives (tests/test_host.py)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ctrl-adapter_amd", "csrc")
NPH = 24


def first_read(i):
    """phase (chunk-relative, may be >= 24 for pieces of the next chunk) of the first read of piece i"""
    if i >= 32:
        return NPH + first_read(i - 32)
    return 2 * (i // 3) if i < 24 else 16 + 4 * ((i - 24) // 4)


def last_read(i):
    if i >= 32:
        return NPH + last_read(i - 32)
    return 2 * (i // 3) + 1 if i < 24 else 16 + 4 * ((i - 24) // 4) + 3


# ---- schedules: {phase: [pieces]} for pieces 8 .. 39 (phase 24 + p = phase p of the next chunk: the loop body issues those pieces in
#      its phase p, for the chunk it is in, when it is not the first chunk) ----
def bulk():
    """round-6 first build: a k-tile's three pieces together, one phase after the tile two k-tiles back was read"""
    s = {}
    for j in range(8, 40):
        s.setdefault(last_read(j - 8) + 1, []).append(j)
    return s


def spread():
    """the same pieces, at most two per phase where the budgets allow it: the DMA instructions sit in the load segment, which the other
    wave row's MFMAs wait for"""
    s = {}
    for t in range(1, 8):
        a, b, c = 3 * t + 5, 3 * t + 6, 3 * t + 7
        s.setdefault(2 * t, []).extend([a, b])
        s.setdefault(2 * t + 1, []).append(c)
    s.setdefault(16, []).extend([29, 30])
    s.setdefault(17, []).append(31)
    s.setdefault(20, []).extend([32, 33])
    s.setdefault(21, []).append(34)
    s.setdefault(22, []).append(35)
    s.setdefault(24, []).extend([36, 37])
    s.setdefault(25, []).extend([38, 39])
    return s


SCHEDULES = {"bulk": bulk, "spread": spread}


def check(sched):
    issue = {}
    for p, ps in sched.items():
        for j in ps:
            assert j not in issue, "piece %d issued twice" % j
            issue[j] = p
    assert sorted(issue) == list(range(8, 40)), "every piece 8 .. 39 exactly once"
    for j, p in issue.items():
        assert p >= last_read(j - 8) + 1, "WAR: piece %d issued in phase %d, piece %d is read until phase %d" % (j, p, j - 8, last_read(j - 8))
        assert p <= first_read(j) - 1, "RAW: piece %d issued in phase %d, first read in phase %d" % (j, p, first_read(j))
    # in-order issue per wave: the counted wait assumes pieces land in stream order
    order = [j for p in sorted(sched) for j in sched[p]]
    assert order == sorted(order), "pieces must be issued in stream order (in-order vmcnt): %s" % order
    return issue


def vmcnt(sched, issue, w):
    """outstanding loads allowed at the wait of phase w (chunk-relative; steady state: the pieces 32.. of the previous chunk's phases
    count as this chunk's 0 .. 7)"""
    need = w + 1
    n = 0
    for j, p in issue.items():
        # this chunk's pieces j < 32 issued in phases <= w, and this chunk's pieces 0 .. 7 issued as 32 .. 39 of the previous chunk
        for jj, pp in ((j, p), (j - 32, p - NPH)):
            if 0 <= jj < 40 and pp <= w and first_read(jj) > need:
                n += 2
    # pieces 32.. issued in this chunk's phases <= w
    return n


def wait_phases():
    return sorted({first_read(i) - 1 for i in range(3, 32)} | {NPH - 1})


def emit(name):
    sched = SCHEDULES[name]()
    issue = check(sched)
    L = []
    L.append("// GENERATED by tools/gen_ffn_schedule.py (schedule \"%s\") -- do not edit; tests/test_host.py checks it against the generator." % name)
    L.append("// Body of the chunk loop of ffn512_kernel: `c` = chunk, `more` = a next chunk exists.  Macros: ffn.hip.")

    def issues(p):
        out = []
        cur = [j for j in sched.get(p, []) if j < 32]
        nxt = [j - 32 for j in sched.get(p, []) if j >= 32]                 # next chunk's pieces, issued from this chunk
        prv = [j - 32 for j in sched.get(p + NPH, []) if j >= 32]           # this chunk's pieces 0 .. 7 issued in ITS OWN phase p
        if cur:
            out.append(" ".join("FF_ISSUE_L(%d, c);" % j for j in cur))
        if nxt:
            out.append("if (more) { " + " ".join("FF_ISSUE_L(%d, c + 1);" % j for j in nxt) + " }")
        if prv:
            out.append("if (c > 0) { " + " ".join("FF_ISSUE_L(%d, c);" % j for j in prv) + " }")
        return " ".join(out)

    waits = {}
    for w in wait_phases():
        waits[w] = vmcnt(sched, issue, w)

    def wait(p):
        if p not in waits:
            return ""
        if p == NPH - 1:
            return "if (more) { FF_VMCNT(%d); }" % waits[p]
        return "FF_VMCNT(%d);" % waits[p]

    def head(p, read):
        iss, wt = issues(p), wait(p)
        s = "        " + read
        if iss:
            s += " " + iss
        if wt:
            s += " " + wt
        s += " FF_PRE();"
        return s

    L.append("        // ================= G1: phases 0 .. 15 (k-tile t = phases 2t, 2t + 1) =================")
    for p in range(16):
        L.append(head(p, "FF_G1_READ(%d, %d);" % (p // 2, p % 2)) + " FF_G1_MMA(); FF_POST();")
    L.append("        // ================= GEGLU of k-steps 0 and 1 (the hidden units 0-15 of every wave column: the exposed half; it frees half of S),")
    L.append("        //                   published to the wave row by one more barrier =================")
    L.append("        jitter();")
    L.append("        FF_GEGLU(0, 0); FF_GEGLU(1, 0); __builtin_amdgcn_sched_barrier(0);")
    L.append("        FF_GEGLU(0, 1); FF_GEGLU(1, 1); __builtin_amdgcn_sched_barrier(0);")
    L.append("        FF_GEGLU(0, 2); FF_GEGLU(1, 2); __builtin_amdgcn_sched_barrier(0);")
    L.append("        FF_GEGLU(0, 3); FF_GEGLU(1, 3);")
    L.append("        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");")
    L.append("        __builtin_amdgcn_sched_barrier(0);")
    L.append("        __builtin_amdgcn_s_barrier();")
    L.append("        __builtin_amdgcn_sched_barrier(0);")
    L.append("        // ================= G2: phases 16 .. 23 (k-step kap = phases 16 + 2 kap, + 1).  The GEGLU of k-steps 2 / 3 runs between the MFMAs of")
    L.append("        //                   k-steps 0 / 1 and goes into the P slot that k-step has just read (its reads retired before the phase's barrier) ======")
    for p in range(16, 24):
        kap, nh = (p - 16) // 2, (p - 16) % 2
        s = head(p, "FF_G2_READ(%d, %d);" % (kap, nh))
        if kap < 2:
            g = kap + 2
            s += " FF_G2_MMA(%d, 0); FF_GEGLU(%d, %d); FF_G2_MMA(%d, 1); FF_GEGLU(%d, %d); FF_G2_MMA(%d, 2); FF_G2_MMA(%d, 3); FF_POST_P();" % (nh, g, 2 * nh, nh, g, 2 * nh + 1, nh, nh)
        elif p == 23:
            s += " FF_G2_MMA(1, 0); FF_G2_MMA(1, 1); s_init(more ? c + 1 : c); FF_G2_MMA(1, 2); FF_G2_MMA(1, 3); FF_POST();"
        else:
            s += " FF_G2_MMA(%d, 0); FF_G2_MMA(%d, 1); FF_G2_MMA(%d, 2); FF_G2_MMA(%d, 3); FF_POST();" % (nh, nh, nh, nh)
        L.append(s)
    return "\n".join(L) + "\n", waits


def main():
    bad = False
    for name in SCHEDULES:
        text, waits = emit(name)
        path = os.path.join(CSRC, "ffn_body_%s.inc" % name)
        if "--check" in sys.argv:
            if not os.path.exists(path) or open(path).read() != text:
                print("%s is not what tools/gen_ffn_schedule.py generates" % path)
                bad = True
        else:
            with open(path, "w") as fh:
                fh.write(text)
            print(name, "vmcnt at the wait phases:", waits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
