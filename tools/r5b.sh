mkdir -p gpurun_out/r5b; O=gpurun_out/r5b
(python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -s -x 2>&1 | grep -E "PARITY qkv one|PARITY groupnorm fused|passed|failed|Error|assert|rror:" | tail -40) > $O/ops.log
(python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -s -k "grouped or golden or gamma_outliers or gain2 or bf16_boundary_at_shape or controlled_step or text_kv or full_size_sdxl or video_chain_at_benched" 2>&1 | grep -E "PARITY|passed|failed|Error|assert|rror:" | tail -60) > $O/e2e.log
python bench.py --per-kernel-out $O/pk.json > $O/bench.json 2> $O/bench.err
for f in none 128x128x32 128x64x64 64x64x64 128x128x64 wide; do
  if [ $f = none ]; then env -u CTRL_IGEMM_FORCE tools/bin/gemm_order_bench $O/small_$f.txt small > /dev/null 2>&1; else CTRL_IGEMM_FORCE=$f tools/bin/gemm_order_bench $O/small_$f.txt small > /dev/null 2>&1; fi
done
tail -c 600 $O/bench.json; cat $O/ops.log | tail -15; cat $O/e2e.log | tail -40
