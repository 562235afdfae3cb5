#!/bin/bash
# the default bench (C5) on several builds of the library: usage gpu_fill_variants.sh <a.so> <b.so> ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for lib in kai-scheduler_amd/csrc/libkai_core.so "$@" kai-scheduler_amd/csrc/libkai_core.so; do
  echo "== $lib"
  KAI_CORE_LIB=$lib KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | grep -E "kai batch|^\{" | sed -e 's/^{.*"ms_per_step": \([0-9.]*\).*"equal_to_oracle": \([a-z]*\).*/ms_per_step \1 equal_to_oracle \2/' | cut -c1-200 | tail -2
done
