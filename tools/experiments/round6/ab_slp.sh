#!/bin/bash
# Same-box A/B of the SLP vectoriser on the attention loops and the fused feed-forward (ctrl-adapter_amd/build.py EXTRA_FLAGS): the tree's
# build (-fno-slp-vectorize on those three files) against CTRL_BUILD_SLP=1 rebuilt ON THE BOX.  One gpurun call.
O=gpurun_out/slp; mkdir -p $O
run() {   # $1 = tag
  timeout 200 tools/bin/attn_bench $O/attn_$1.txt 0,2,3,12 > /dev/null 2>&1
  timeout 120 tools/bin/ffn_bench $O/ffn_$1.txt > /dev/null 2>&1
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/bench_$1_$i.json; done
  timeout 300 python bench.py --workload svd16 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/bench_svd_$1.json
}
run noslp
CTRL_BUILD_SLP=1 CTRL_BUILD_AUDIT=0 python ctrl-adapter_amd/build.py | tail -1
run slp
CTRL_BUILD_AUDIT=0 python ctrl-adapter_amd/build.py | tail -1
run noslp2
for t in noslp slp noslp2; do echo "== $t"; grep -E "^B|variant" $O/attn_$t.txt | grep -v host | cut -c1-60; grep " M " $O/ffn_$t.txt | cut -c1-80
  for f in $O/bench_${t}_1.json $O/bench_${t}_2.json $O/bench_svd_$t.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print(sys.argv[1], d["ms_per_step"], (d.get("fused_step") or {}).get("ms_per_step"), d["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
  done
done
