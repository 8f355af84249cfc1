#!/bin/bash
# Same-box A/B of the row maximum of the attention loops: v_maximum3_f32 (__builtin_elementwise_maximum, the tree's build) against fmaxf
# (CTRL_BUILD_FMAXF=1 rebuilt on the box).  Bit-identical results; one gpurun call.
O=gpurun_out/fmax; mkdir -p $O
run() {
  timeout 200 tools/bin/attn_bench $O/attn_$1.txt 0,2 > /dev/null 2>&1
  for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > $O/bench_$1_$i.json; done
}
run new
CTRL_BUILD_FMAXF=1 CTRL_BUILD_AUDIT=0 python ctrl-adapter_amd/build.py | tail -1
run fmaxf
CTRL_BUILD_AUDIT=0 python ctrl-adapter_amd/build.py | tail -1
run new2
for t in new fmaxf new2; do echo "== $t"; grep -E "^B|variant" $O/attn_$t.txt | grep -v host | cut -c1-60
  for f in $O/bench_${t}_1.json $O/bench_${t}_2.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); print(sys.argv[1], d["ms_per_step"], (d.get("fused_step") or {}).get("ms_per_step"), d["roofline"]["avg_launch_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
  done
done
