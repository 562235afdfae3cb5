#!/bin/bash
# round-3: victim search profile on 32 workgroups + the oracle on the same box's host cores.  usage: gpu_r3e.sh <tag>
TAG=${1:-r03e}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "== C4 10% on the default number of workgroups, with the oracle (cpu_baseline) timed on this box"
  KAI_PROF=1 timeout 1200 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 2>&1 | grep -E "kai victim|^\{"
  echo "== C4 10%, phase clocks of workgroup 0 (-DKAI_PROF_VICTIM)"
  KAI_CORE_LIB=build/libkai_core_vprof.so KAI_PROF=1 timeout 900 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "kai prof|kai victim" | cut -c1-1500
} > gpurun_out/${TAG}_c4.txt 2>&1
cut -c1-1800 gpurun_out/${TAG}_c4.txt
