mkdir -p gpurun_out/r5d; O=gpurun_out/r5d
(python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -s -k "groupnorm or conv3x3_small or qkv_one" 2>&1 | grep -E "PARITY groupnorm fused|passed|failed|FAILED|Error|assert|rror:" | tail -30) > $O/ops.log
B="python bench.py --no-other-workloads --no-cpu-baseline --steps 20 --warmup 5"
$B --per-kernel-out $O/pk_default.json > $O/b_default.json 2>> $O/bench.err
CTRL_GN_FUSED=0 $B --per-kernel-out $O/pk_tmp.json > $O/b_nognfused.json 2>> $O/bench.err
$B --per-kernel-out $O/pk_tmp.json > $O/b_default2.json 2>> $O/bench.err
for f in default nognfused default2; do python - <<PY
import json
try:
    d=json.loads(open("$O/b_$f.json").read().strip().splitlines()[-1])
    print("%-14s ms/step %.3f median %s fused %s launches %s" % ("$f", d["ms_per_step"], d.get("ms_per_step_median"), (d.get("fused_step") or {}).get("ms_per_step"), d.get("launches_per_step")))
except Exception as e:
    print("$f", "FAILED", e)
PY
done | tee $O/ab.txt
cat $O/ops.log | tail -12
python - <<PY
import json
pk=json.load(open("$O/pk_default.json"))
print({k:(v['launches_per_step'],v['ms_per_step']) for k,v in pk['kernels'].items()})
print([ (r['kernel'][:40], r['bound']) for r in pk['per_kernel'] if r['bound'] is None])
PY
