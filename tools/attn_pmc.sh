#!/bin/bash
# Issue-slot counters of the L = 16384 self-attention kernel, plain and folded form (one rocprofv3 --pmc pass each, nothing
# but the counter collection):  bash tools/attn_pmc.sh  -> gpurun_out/attn_pmc/{attn,attn_fold}.csv summary lines
set -u
O=gpurun_out/attn_pmc; mkdir -p $O; export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"
for w in attn attn_fold; do
  rocprofv3 --pmc $C --output-format csv -d $O/$w -- python tools/one_kernel.py $w > $O/$w.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$O/$w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flash_attn" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
launches = 3.0
print("$w", {k: round(v / launches / 1e6, 1) for k, v in sorted(acc.items())}, "(millions per launch)")
wc = acc.get("SQ_WAVE_CYCLES", 0)
if wc:
    print("   fractions of SQ_WAVE_CYCLES:", {k: round(acc[k] / wc, 3) for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if k in acc})
PY
done 2>&1 | tee $O/summary.txt
