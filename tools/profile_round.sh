#!/bin/bash
# Regenerates the measurement artefacts of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02 v1 [1 = skip the pytest parity logs]
# Writes under gpurun_out/final/; copy what should be judged into profiles/.
set -u
R=${1:-r01}; V=${2:-v4}; SKIPTESTS=${3:-0}
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
if [ "$SKIPTESTS" = "0" ]; then
(python -m pytest tests/test_gpu_ops.py tests/test_image_prep.py -m gpu -q --tb=short -s 2>&1 | grep -E "PARITY|passed|failed|Error") > $O/${R}_op_parity_${V}.log
(python -m pytest tests/test_gpu_e2e.py -m gpu -q --tb=short -s 2>&1 | grep -E "PARITY|passed|failed|Error") > $O/${R}_e2e_parity_${V}.log
fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/${R}_smoke_${V}.log 2>&1
python bench.py > $O/${R}_bench_${V}.json 2> $O/bench.err
python bench.py --workload svd16 --steps 10 --warmup 3 > $O/${R}_bench_svd16_${V}.json 2>> $O/bench.err
python bench.py --workload i2vgen16 --steps 10 --warmup 3 --no-cpu-baseline > $O/${R}_bench_i2vgen16_${V}.json 2>> $O/bench.err
python bench.py --workload multi3 --steps 10 --warmup 3 --no-cpu-baseline > $O/${R}_bench_multi3_${V}.json 2>> $O/bench.err
timeout 300 python tools/microbench.py > $O/${R}_microbench_${V}.log 2>&1
# per-kernel time of the same command (its own run: no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${R}_rocprofv3_kernel_stats_${V}.csv
# the same with the stream lanes OFF and eager launches: kernels do not overlap, so per-symbol durations are the ones
# bench.py's HIP events see (roofline.avg_launch_ms); summarised per (symbol, grid) + the per-launch dump of the library
CTRL_ADAPTER_LANES=1 CTRL_CN_AUX=0 CTRL_PROF_DUMP=$O/${R}_launches_${V}.tsv rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lanes_off -- \
  python bench.py --no-graph --no-cpu-baseline --steps 3 --warmup 1 > $O/stats_lanes_off.log 2>&1
cp $(ls $O/stats_lanes_off/*/*kernel_stats.csv | head -1) $O/${R}_rocprofv3_kernel_stats_lanes_off_${V}.csv
LSTEPS=$(grep -h avgpool $O/stats_lanes_off/*/*kernel_trace.csv | wc -l)
python tools/kernel_trace_summary.py $O/stats_lanes_off $LSTEPS $O/${R}_rocprofv3_per_kernel_lanes_off_${V}.csv
rm -f $O/stats_lanes_off/*/*kernel_trace.csv
# HBM traffic: one counter per pass, nothing but the counter collection; one stream so that counters attribute cleanly
for c in FETCH_SIZE WRITE_SIZE; do
  CTRL_ADAPTER_LANES=1 CTRL_CN_AUX=0 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --no-graph --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
done
# warmup 1 + timed 2 + 3 profiled steps + 2 eager steps before capture are all counted: steps = number of avgpool launches
STEPS=$(grep -h avgpool $O/pmc_FETCH_SIZE/*/*counter_collection.csv | wc -l)
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $STEPS $O/${R}_pmc_hbm_traffic_${V}.json $O/${R}_launches_${V}.tsv
rm -f $O/stats/*/*kernel_trace.csv
tail -n 2 $O/${R}_op_parity_${V}.log $O/${R}_e2e_parity_${V}.log $O/${R}_smoke_${V}.log; cat $O/${R}_bench_${V}.json | head -c 600
