#!/bin/bash
# Issue-slot counters of the L = 16384 self-attention kernel for a list of ctrl_attn_set_variant numbers, through the torch-free
# harness (one rocprofv3 --pmc pass per variant, nothing but the counter collection):
#   bash tools/attn_pmc.sh "0 2 9"   -> gpurun_out/attn_pmc/summary.txt
set -u
O=gpurun_out/attn_pmc; mkdir -p $O; export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"
# (the eight counters of round 2's pass: a ninth one made rocprofv3 abort on this pool; every pass under its own timeout)
for v in ${1:-0 2}; do
  for pass in 1; do
    timeout 90 rocprofv3 --pmc $C --output-format csv -d $O/v${v}_p$pass -- tools/bin/attn_bench $O/v${v}_p$pass.txt $v 1 > $O/v${v}_p$pass.log 2>&1
  done
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$O/v${v}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flash_attn" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
launches = max(n.values()) if n else 1
print("variant $v  (%d launches)" % launches, {k: round(val / launches / 1e6, 1) for k, val in sorted(acc.items())}, "(millions per launch)")
wc = acc.get("SQ_WAVE_CYCLES", 0)
if wc:
    print("   fractions of SQ_WAVE_CYCLES:", {k: round(acc[k] / wc, 3) for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if k in acc})
PY
done 2>&1 | tee $O/summary.txt
