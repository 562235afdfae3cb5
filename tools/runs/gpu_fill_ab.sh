#!/bin/bash
# A/B of two builds of the library on the default bench (C5): usage gpu_fill_ab.sh <variant.so>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for lib in kai-scheduler_amd/csrc/libkai_core.so $1 kai-scheduler_amd/csrc/libkai_core.so $1; do
  echo "== $lib"
  KAI_CORE_LIB=$lib KAI_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | grep -E "kai batch|^\{" | sed -e 's/^{.*"ms_per_step": \([0-9.]*\).*"equal_to_oracle": \([a-z]*\).*/ms_per_step \1 equal_to_oracle \2/' | tail -2
done
