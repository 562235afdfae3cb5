#!/bin/bash
# victim-search profile: C4 at the given scales, per-action phase clocks (library built with -DKAI_PROF_VICTIM into build/libkai_core_vprof.so)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for sc in ${@:-0.03}; do
  echo "== C4 scale $sc"
  KAI_CORE_LIB=build/libkai_core_vprof.so KAI_PROF=1 timeout 900 python bench.py --config C4 --scale $sc --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "kai prof|kai batch|^\{" | cut -c1-1200
done
