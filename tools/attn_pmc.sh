#!/bin/bash
# Issue-slot, LDS and HBM counters of the head_dim-64 self-attention kernel (csrc/attention_d64.hip, the default variant 2) and of the round-2
# kernel (variant 0) beside it, through the torch-free harness tools/bin/attn_bench on its first shape (one rocprofv3 --pmc pass per counter set):
#   bash tools/attn_pmc.sh [tag]   -> gpurun_out/attn_pmc/summary_<tag>.txt
set -u
T=${1:-v0}
O=gpurun_out/attn_pmc; mkdir -p $O; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
rm -rf $O/issue $O/lds $O/fetch $O/write
timeout 120 rocprofv3 --pmc $A --output-format csv -d $O/issue -- tools/bin/attn_bench /dev/null 2,0 1 > $O/issue.log 2>&1
timeout 120 rocprofv3 --pmc $B --output-format csv -d $O/lds -- tools/bin/attn_bench /dev/null 2,0 1 > $O/lds.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- tools/bin/attn_bench /dev/null 2,0 1 > $O/fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- tools/bin/attn_bench /dev/null 2,0 1 > $O/write.log 2>&1
python3 - <<PY 2>&1 | tee $O/summary_$T.txt
import csv, glob, collections
def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
            if "flash_attn" not in name: continue
            k = (name.split("((")[0][:48], int(r["Grid_Size"]))
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return acc, n
ia, inn = collect("$O/issue"); la, ln = collect("$O/lds"); fa, fn = collect("$O/fetch"); wa, wn = collect("$O/write")
print("per (kernel, grid), averages per launch.  Counters in millions; fractions of SQ_WAVE_CYCLES (quad-cycles summed over waves);")
print("MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE); HBM bytes = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, KB units")
for k in sorted(ia, key=lambda k: -ia[k].get("SQ_WAVE_CYCLES", 0)):
    a = ia[k]; L = max(inn[k].values()); wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    row = {c.replace("SQ_", ""): round(a[c] / L / 1e6, 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES") if c in a}
    fr = {c.replace("SQ_", ""): round(a[c] / wc, 3) for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if c in a}
    extra = ""
    if k in la and la[k].get("SQ_LDS_IDX_ACTIVE"):
        l = la[k]; Ll = max(ln[k].values())
        gui = l.get("GRBM_GUI_ACTIVE", 0) / Ll
        extra = "  LDS conflict/active %.3f  LDS active / kernel cycles / CU %.3f  WAIT_INST_LDS/WAVE_CYCLES %.3f  kernel cycles %.0fk" % (
            l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"], l["SQ_LDS_IDX_ACTIVE"] / Ll / 256.0 / max(gui, 1), l.get("SQ_WAIT_INST_LDS", 0) / max(l.get("SQ_WAVE_CYCLES", 1), 1), gui / 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a and gui:
            extra += "  MFMA busy %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / L / (1024.0 * gui))
    hb = ""
    if k in fa and k in wa:
        f = fa[k]["FETCH_SIZE"] / max(fn[k]["FETCH_SIZE"], 1); w = wa[k]["WRITE_SIZE"] / max(wn[k]["WRITE_SIZE"], 1)
        hb = "  HBM read %.1f MB (x2 corrected) write %.1f MB" % (f * 2 * 1024 / 1e6, w * 1024 / 1e6)
    print("%-60s grid %8d x%d\n      %s %s%s%s" % (k[0], k[1], L, row, fr, extra, hb))
PY
