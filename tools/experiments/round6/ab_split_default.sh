#!/bin/bash
# Same-box A/B of the ControlNet's per-conv split default (CTRL_CN_SPLIT_RESNET_LEVELS 3 = round 5, 1 = round 6): SDXL b = 8 twice each,
# alternating, then SVD-16 and the three-net workload once each.  One gpurun call; prints ms/step, value and the fused-step time.
mkdir -p gpurun_out
for rl in 3 1 3 1; do CTRL_CN_SPLIT_RESNET_LEVELS=$rl timeout 300 python bench.py --no-cpu-baseline --no-other-workloads 2>gpurun_out/r06_rl_err.txt | tail -1 > gpurun_out/r06_rl${rl}_$RANDOM.json; done
for rl in 3 1; do
  CTRL_CN_SPLIT_RESNET_LEVELS=$rl timeout 300 python bench.py --workload svd16 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r06_rl_svd_${rl}.json
  CTRL_CN_SPLIT_RESNET_LEVELS=$rl timeout 300 python bench.py --workload multi3 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 > gpurun_out/r06_rl_multi3_${rl}.json
done
for f in gpurun_out/r06_rl*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(sys.argv[1], d["ms_per_step"], d["value"], (d.get("fused_step") or {}).get("ms_per_step"), d.get("launches_per_step"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -3 gpurun_out/r06_rl_err.txt
