#!/bin/bash
# rocprofv3 evidence for the bench workload: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes.
# usage: gpu_prof.sh <tag> [bench args...]
TAG=${1:-x}; shift; R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; cd /tmp && export TMPDIR=/tmp
ARGS="${@:---config C5 --steps 1 --warmup 1 --cpu-sample 0}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_${TAG}_trace" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/prof_${TAG}_trace.log" 2>&1; echo "trace rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/prof_${TAG}_$ctr" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/prof_${TAG}_$ctr.log" 2>&1; echo "$ctr rc=$?"
done
find "$R/gpurun_out" -path "*prof_${TAG}*" -name '*.csv' | head -20
