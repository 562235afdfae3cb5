#!/bin/bash
# PMC FETCH_SIZE / WRITE_SIZE passes (their own runs, kernel trace only) of the mixed config 5 on the sequential engine + scan grid.  usage: gpu_pmc_mixed.sh <tag>
TAG=${1:-r03y}; R="$GRAFT_REPO_ROOT"; cd /tmp; export TMPDIR=/tmp; mkdir -p "$R/gpurun_out"
ARGS="--config C5 --mixed --steps 1 --warmup 0 --cpu-sample 0"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_mixed_$ctr" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/${TAG}_prof_mixed_$ctr.log" 2>&1; echo "$ctr rc=$?"
done
