#!/bin/bash
# Side build of the library for experiments that must not touch the product .so: copies csrc/ + include/ + the harnesses to
# tools/bin/alt/ (git-ignored, travels to the GPU box), applies the experiment patches of this directory, builds
# tools/bin/alt/libctrlhip.so and the harnesses linked against IT.
#   bash tools/experiments/build_side_library.sh            (from the repo root; ~1 minute)
#   gpurun -- 'tools/bin/alt/attn_bench_alt out.txt 2,15,16,19,21,24,25,26 1'     # ablations / ping-pong / priorities
#   gpurun -- 'tools/bin/alt/gemm_order_bench_nt out.txt nt'  |  '... data'         # nt hints / random vs zero operands
#   gpurun -- 'tools/bin/alt/occ_probe out.txt'                                      # occupancy + data dependence of attention
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
ALT="$ROOT/tools/bin/alt"
rm -rf "$ALT"; mkdir -p "$ALT/build"
cp -r "$ROOT/ctrl-adapter_amd/csrc" "$ALT/csrc"
cp -r "$ROOT/include" "$ALT/include"
cp "$ROOT/tools/gemm_order_bench.cpp" "$ROOT/tools/attn_bench.cpp" "$ROOT/tools/experiments/occ_probe.cpp" "$ALT/"
for p in nt_hint.patch attention_ablation_pingpong.patch; do
  sed 's#ctrl-adapter_amd/csrc/#csrc/#g; s#tools/gemm_order_bench.cpp#gemm_order_bench.cpp#g' "$ROOT/tools/experiments/$p" | patch -s -p0 -d "$ALT"
done
sed -i 's#"../../include/ctrl_hip.h"#"../include/ctrl_hip.h"#' "$ALT"/csrc/*.h "$ALT"/csrc/*.cpp "$ALT"/csrc/*.hip
cd "$ALT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -w -Iinclude"
pids=()
for f in csrc/*.hip csrc/*.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o build/$(basename $f).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libctrlhip.so build/*.o
for h in gemm_order_bench:gemm_order_bench_nt attn_bench:attn_bench_alt occ_probe:occ_probe; do
  /opt/rocm/bin/hipcc -O2 -std=c++17 -w -Iinclude ${h%%:*}.cpp -o ${h##*:} -L. -lctrlhip -Wl,-rpath,'$ORIGIN'
done
echo "side build ready: $ALT"
