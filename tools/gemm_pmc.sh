#!/bin/bash
# Issue-slot and LDS counters of the implicit-GEMM kernels on the path's shapes, through the torch-free harness
# (tools/gemm_order_bench shapes: 16 shapes with their real epilogues; one rocprofv3 --pmc pass per counter set, nothing but the
# counter collection, every pass under its own timeout):
#   bash tools/gemm_pmc.sh   -> gpurun_out/gemm_pmc/summary.txt
set -u
O=gpurun_out/gemm_pmc; mkdir -p $O; export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
timeout 90 rocprofv3 --pmc $A --output-format csv -d $O/issue -- tools/bin/gemm_order_bench /dev/null shapes > $O/issue.log 2>&1
timeout 90 rocprofv3 --pmc $B --output-format csv -d $O/lds -- tools/bin/gemm_order_bench /dev/null shapes > $O/lds.log 2>&1
python3 - <<PY 2>&1 | tee $O/summary.txt
import csv, glob, collections, re, sys
sys.path.insert(0, "tools")
from pmc_traffic import symbol_of, kernel_class
def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_class(r["Kernel_Name"]) is None: continue
            k = (symbol_of(r["Kernel_Name"]), int(r["Grid_Size"]))
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return acc, n
ia, inn = collect("$O/issue"); la, ln = collect("$O/lds")
print("per (kernel, grid): millions per launch; fractions of SQ_WAVE_CYCLES; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles) is left to the reader")
for k in sorted(ia, key=lambda k: -ia[k].get("SQ_WAVE_CYCLES", 0)):
    a = ia[k]; L = max(inn[k].values()); wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    row = {c: round(a[c] / L / 1e6, 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES") if c in a}
    fr = {c: round(a[c] / wc, 3) for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if c in a}
    extra = ""
    if k in la and la[k].get("SQ_LDS_IDX_ACTIVE"):
        extra = "  LDS bank conflict / LDS active = %.3f" % (la[k].get("SQ_LDS_BANK_CONFLICT", 0.0) / la[k]["SQ_LDS_IDX_ACTIVE"])
    print("%-52s grid %8d  x%d  %s  %s%s" % (k[0], k[1], L, row, fr, extra))
PY
