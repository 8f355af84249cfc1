#!/bin/bash
# Same-box A/B of the uniform wave index in the attention kernels (wave = readfirstlane(tid >> 6): LDS-DMA destinations and per-wave offsets in
# SGPRs, 5 VALU per tile fewer): the tree's build, then the old line put back with sed and rebuilt ON THE BOX.  Bit-identical results.
O=gpurun_out/wave; mkdir -p $O
run() {
  timeout 200 tools/bin/attn_bench $O/attn_$1.txt 0,2 > /dev/null 2>&1
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/bench_$1_$i.json; done
}
run new
sed -i 's/wave = __builtin_amdgcn_readfirstlane(tid >> 6);/wave = tid >> 6;/' ctrl-adapter_amd/csrc/attention_d64.hip ctrl-adapter_amd/csrc/attention.hip
CTRL_BUILD_AUDIT=0 python ctrl-adapter_amd/build.py | tail -1
run old
for t in new old; do echo "== $t"; grep -E "^B|variant" $O/attn_$t.txt | grep -v host | cut -c1-60
  for f in $O/bench_${t}_1.json $O/bench_${t}_2.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print(sys.argv[1], d["ms_per_step"], (d.get("fused_step") or {}).get("ms_per_step"), d["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
  done
done
