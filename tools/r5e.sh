mkdir -p gpurun_out/r5e; O=gpurun_out/r5e
(python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -s -k "controlnet_golden or text_kv or condition_cache or full_size_sdxl or controlled_step or output_container or global_pool or boundary_dtypes" 2>&1 | grep -E "PARITY|passed|failed|FAILED|Error|assert|rror:" | tail -40) > $O/e2e.log
B="python bench.py --no-other-workloads --no-cpu-baseline --steps 20 --warmup 5"
$B --per-kernel-out $O/pk_default.json > $O/b_default.json 2>> $O/bench.err
CTRL_GROUP=0 $B --per-kernel-out $O/pk_tmp.json > $O/b_nogroup.json 2>> $O/bench.err
for f in default nogroup; do python - <<PY
import json
try:
    d=json.loads(open("$O/b_$f.json").read().strip().splitlines()[-1])
    print("%-14s ms/step %.3f median %s fused %s launches %s" % ("$f", d["ms_per_step"], d.get("ms_per_step_median"), (d.get("fused_step") or {}).get("ms_per_step"), d.get("launches_per_step")))
except Exception as e:
    print("$f", "FAILED", e)
PY
done | tee $O/ab.txt
for f in none 128x128x32 128x256 256x128 wide; do
  if [ $f = none ]; then env -u CTRL_IGEMM_FORCE tools/bin/gemm_order_bench $O/shapes_$f.txt shapes > /dev/null 2>&1; else CTRL_IGEMM_FORCE=$f tools/bin/gemm_order_bench $O/shapes_$f.txt shapes > /dev/null 2>&1; fi
done
cat $O/e2e.log | tail -30
