#!/bin/bash
# topology-path profile of the allocate action: mixed C5 at the given scales, per-phase control-lane clocks (library built with -DKAI_PROF_VICTIM)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for sc in ${@:-0.25}; do
  echo "== C5 mixed scale $sc"
  KAI_CORE_LIB=build/libkai_core_vprof.so KAI_PROF=1 timeout 600 python bench.py --config C5 --mixed --scale $sc --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "kai prof|kai batch|^\{" | cut -c1-1500
done
