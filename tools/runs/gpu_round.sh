#!/bin/bash
# End-of-milestone GPU evidence: smoke(), the default bench line, the per-kernel summary and the PMC traffic.  usage: gpu_round.sh <tag>
TAG=${1:-x}; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as G; G.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke_$TAG.log
timeout 600 python bench.py > gpurun_out/bench_default_$TAG.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/bench_default_$TAG.log | cut -c1-400
bash tools/gpu_prof.sh $TAG
