#!/bin/bash
# Regenerates the measurement artefacts of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r06 v1 [1 = also write the pytest parity logs (adds ~13 minutes; they run LAST)]
# Writes under gpurun_out/final/; copy what should be judged into profiles/.  ~8 minutes of box time without the tests.
set -u
R=${1:-r05}; V=${2:-v1}; TESTS=${3:-0}
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
if [ "${ONLY_TRACES:-0}" != "1" ]; then      # ONLY_TRACES=1: just the rocprofv3 passes below
python -c "import __graft_entry__ as g; g.smoke()" > $O/${R}_smoke_${V}.log 2>&1
# the headline exactly as the driver runs it, then the other BASELINE configurations (no CPU leg: it is the slow part)
python bench.py --per-kernel-out $O/${R}_per_kernel_${V}.json > $O/${R}_bench_${V}.json 2> $O/bench.err      # (carries other_workloads: svd16, i2vgen16, multi3, sdxl b16)
python bench.py --workload svd16 --steps 10 --warmup 3 --no-cpu-baseline --per-kernel-out $O/${R}_per_kernel_svd16_${V}.json > $O/${R}_bench_svd16_${V}.json 2>> $O/bench.err
python bench.py --workload i2vgen16 --steps 10 --warmup 3 --no-cpu-baseline --per-kernel-out $O/${R}_per_kernel_i2vgen16_${V}.json > $O/${R}_bench_i2vgen16_${V}.json 2>> $O/bench.err
python bench.py --workload multi3 --steps 10 --warmup 3 --no-cpu-baseline --per-kernel-out $O/${R}_per_kernel_multi3_${V}.json > $O/${R}_bench_multi3_${V}.json 2>> $O/bench.err
python bench.py --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --per-kernel-out $O/${R}_per_kernel_b16_${V}.json > $O/${R}_bench_b16_${V}.json 2>> $O/bench.err
python bench.py --workload i2vgen16 --clip-split --steps 5 --warmup 2 --no-cpu-baseline --per-kernel-out $O/${R}_per_kernel_clip_${V}.json > $O/${R}_bench_clip_split_world1_${V}.json 2>> $O/bench.err
timeout 120 tools/bin/gemm_order_bench $O/${R}_gemm_path_shapes_${V}.txt shapes > /dev/null 2>&1
timeout 120 tools/bin/attn_bench $O/${R}_attention_variants_${V}.txt > /dev/null 2>&1
timeout 120 tools/bin/ffn_bench $O/${R}_ffn_bench_${V}.txt > /dev/null 2>&1
# the ControlNet call alone, per kernel (round 6)
python bench.py --no-cpu-baseline --no-other-workloads --steps 5 --warmup 2 --profile-scope controlnet --per-kernel-out $O/${R}_per_kernel_controlnet_${V}.json > /dev/null 2>> $O/bench.err
# counters of the fused feed-forward kernel and of the two-launch form beside it
bash tools/ffn_pmc.sh ${R}_${V} > /dev/null 2>&1; cp gpurun_out/ffn_pmc/summary_${R}_${V}.txt $O/${R}_ffn_pmc_${V}.txt 2>/dev/null
fi
# per-kernel time of the same command (its own run: no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 --per-kernel-out $O/tmp_pk.json > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${R}_rocprofv3_kernel_stats_${V}.csv
# the same with the stream lanes OFF and eager launches: kernels do not overlap, so per-symbol durations are the ones
# bench.py's HIP events see (roofline.avg_launch_ms); summarised per (symbol, grid) + the per-launch dump of the library
CTRL_ADAPTER_LANES=1 CTRL_CN_AUX=0 CTRL_PROF_DUMP=$O/${R}_launches_${V}.tsv rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lanes_off -- \
  python bench.py --no-graph --no-cpu-baseline --steps 3 --warmup 1 --per-kernel-out $O/tmp_pk.json > $O/stats_lanes_off.log 2>&1
cp $(ls $O/stats_lanes_off/*/*kernel_stats.csv | head -1) $O/${R}_rocprofv3_kernel_stats_lanes_off_${V}.csv
LSTEPS=$(grep -h avgpool $O/stats_lanes_off/*/*kernel_trace.csv | wc -l)
python tools/kernel_trace_summary.py $O/stats_lanes_off $LSTEPS $O/${R}_rocprofv3_per_kernel_lanes_off_${V}.csv
rm -f $O/stats_lanes_off/*/*kernel_trace.csv
# HBM traffic: one counter per pass, nothing but the counter collection; one stream so that counters attribute cleanly
for c in FETCH_SIZE WRITE_SIZE; do
  CTRL_ADAPTER_LANES=1 CTRL_CN_AUX=0 timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --no-graph --no-cpu-baseline --steps 2 --warmup 1 --per-kernel-out $O/tmp_pk.json > $O/pmc_$c.log 2>&1
done
# warmup 1 + timed 2 + 3 profiled steps + 2 eager steps before capture are all counted: steps = number of avgpool launches
STEPS=$(grep -h avgpool $O/pmc_FETCH_SIZE/*/*counter_collection.csv | wc -l)
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $STEPS $O/${R}_pmc_hbm_traffic_${V}.json $O/${R}_launches_${V}.tsv
rm -f $O/stats/*/*kernel_trace.csv $O/tmp_pk.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/stats $O/stats_lanes_off
tail -n 2 $O/${R}_smoke_${V}.log; tail -c 600 $O/${R}_bench_${V}.json
# the parity logs last: a budget cut must not cost the measurements above
if [ "$TESTS" = "1" ]; then
(python -m pytest tests/test_gpu_ops.py tests/test_image_prep.py -m gpu -q --tb=short -s 2>&1 | grep -E "PARITY|passed|failed|FAILED|Error|assert") > $O/${R}_op_parity_${V}.log
(python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -s 2>&1 | grep -E "PARITY|passed|failed|FAILED|Error|assert") > $O/${R}_e2e_parity_${V}.log
fi

