mkdir -p gpurun_out/r5h; O=gpurun_out/r5h
CTRL_ADAPTER_LANES=1 GRP=1 python tools/diag/graph_1lane.py 2>&1 | grep -E "replay|eager" > $O/diag_1lane.txt
GRP=1 python tools/diag/graph_1lane.py 2>&1 | grep -E "replay|eager" > $O/diag_4lane.txt
(python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -s -k "splitk or conv3x3 or groupnorm or layout or split_operand" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -10) > $O/ops.log
(python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -s -k "grouped or golden or multi_condition_router or text_kv or scatter or controlled_step or full_size_sdxl or video_chain_at_benched or weight_distribution or clip_sharded_adapter_equals" 2>&1 | grep -E "PARITY|passed|failed|FAILED|Error|assert" | tail -60) > $O/e2e.log
python bench.py --no-other-workloads --no-cpu-baseline --steps 20 --warmup 5 --per-kernel-out $O/pk.json > $O/bench.json 2> $O/bench.err
cat $O/diag_1lane.txt $O/diag_4lane.txt; cat $O/ops.log; grep -E "config-5 chain|svd16-shape chain|grouped|gamma|passed|failed|FAILED|assert|golden plain" $O/e2e.log | cut -c1-330; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms/step %.3f median %s fused %s launches %s" % (d["ms_per_step"], d.get("ms_per_step_median"), (d.get("fused_step") or {}).get("ms_per_step"), d.get("launches_per_step")))
PY
