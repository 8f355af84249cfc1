#!/bin/bash
# Same-box A/B of LLVM's AMDGPU scheduling strategies on the attention files (CTRL_BUILD_ATTN_SCHED, rebuilt ON THE BOX): default, max-ilp,
# max-memory-clause, iterative-ilp, default again.  tools/bin/attn_bench variants 0 and 2, three shapes.  One gpurun call.
O=gpurun_out/sched; mkdir -p $O
for st in default max-ilp max-memory-clause iterative-ilp default2; do
  if [ $st = default ] || [ $st = default2 ]; then unset CTRL_BUILD_ATTN_SCHED; else export CTRL_BUILD_ATTN_SCHED=$st; fi
  CTRL_BUILD_AUDIT=0 python ctrl-adapter_amd/build.py | tail -1
  timeout 200 tools/bin/attn_bench $O/attn_$st.txt 0,2 > /dev/null 2>&1
  echo "== $st"; grep -E "^B|variant" $O/attn_$st.txt | grep -v host | cut -c1-60
done
