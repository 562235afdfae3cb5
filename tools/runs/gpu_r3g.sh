#!/bin/bash
# round-3: A/B of the fill kernel (runs of same-class tasks per node vs task by task) + the mixed C5 line.  usage: gpu_r3g.sh <tag>
TAG=${1:-r03g}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for lib in kai-scheduler_amd/csrc/libkai_core.so build/libkai_core_noruns.so; do
    echo "== $lib"
    for cfg in C5 C3 C2; do KAI_CORE_LIB=$lib KAI_PROF=1 timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-sample 0 2>&1 | grep -E "kai batch|^\{" | cut -c1-330; done
  done
  echo "== mixed C5 (sequential engine)"
  KAI_PROF=1 timeout 900 python bench.py --config C5 --mixed --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "^\{" | cut -c1-900
} > gpurun_out/${TAG}_fill_ab.txt 2>&1
cat gpurun_out/${TAG}_fill_ab.txt
