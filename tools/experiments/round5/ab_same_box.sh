mkdir -p gpurun_out/r5c; O=gpurun_out/r5c
(python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -s 2>&1 | grep -E "PARITY qkv one|PARITY groupnorm fused|PARITY conv3x3_small|passed|failed|FAILED|Error|assert|rror:" | tail -60) > $O/ops.log
(python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -s -k "grouped or golden or controlled_step or text_kv or full_size_sdxl or sdxl_batch8 or video_chain_at_benched or multi_condition_and_i2vgen" 2>&1 | grep -E "PARITY|passed|failed|FAILED|Error|assert|rror:" | tail -70) > $O/e2e.log
B="python bench.py --no-other-workloads --no-cpu-baseline --steps 20 --warmup 5"
$B --per-kernel-out $O/pk_default.json > $O/b_default.json 2>> $O/bench.err
CTRL_GROUP=0 $B --per-kernel-out $O/pk_nogroup.json > $O/b_nogroup.json 2>> $O/bench.err
CTRL_SMALL_TILES=0 $B --per-kernel-out $O/pk_tmp.json > $O/b_nosmalltiles.json 2>> $O/bench.err
CTRL_SMALLCONV_MFMA=0 $B --per-kernel-out $O/pk_tmp.json > $O/b_nosmallconv.json 2>> $O/bench.err
CTRL_GN_FUSED=0 $B --per-kernel-out $O/pk_tmp.json > $O/b_nognfused.json 2>> $O/bench.err
CTRL_QKV_ONE=0 $B --per-kernel-out $O/pk_tmp.json > $O/b_noqkvone.json 2>> $O/bench.err
$B --workload svd16 --steps 10 --per-kernel-out $O/pk_svd16.json > $O/b_svd16.json 2>> $O/bench.err
CTRL_GROUP=0 CTRL_SMALLCONV_MFMA=0 CTRL_SMALL_TILES=0 CTRL_GN_FUSED=0 CTRL_QKV_ONE=0 $B --workload svd16 --steps 10 --per-kernel-out $O/pk_tmp.json > $O/b_svd16_r4like.json 2>> $O/bench.err
rm -f $O/pk_tmp.json
for f in default nogroup nosmalltiles nosmallconv nognfused noqkvone svd16 svd16_r4like; do python - <<PY
import json
try:
    d=json.loads(open("$O/b_$f.json").read().strip().splitlines()[-1])
    print("%-14s ms/step %.3f median %s fused %s launches %s" % ("$f", d["ms_per_step"], d.get("ms_per_step_median"), (d.get("fused_step") or {}).get("ms_per_step"), d.get("launches_per_step")))
except Exception as e:
    print("$f", "FAILED", e)
PY
done | tee $O/ab.txt
cat $O/ops.log | tail -25; cat $O/e2e.log | tail -45
