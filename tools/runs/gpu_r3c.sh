#!/bin/bash
# round-3: the victim search on several workgroups.  usage: gpu_r3c.sh <tag> [workgroup counts]
TAG=${1:-r03c}; shift; WGS=${@:-64 128}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "several_workgroups or eviction_order or integration or decisions_close" > gpurun_out/${TAG}_pytest_multi.txt 2>&1; echo "pytest multi rc=$?"; tail -5 gpurun_out/${TAG}_pytest_multi.txt
for wgs in $WGS; do
  for sc in 0.03 0.1; do
    echo "== C4 scale $sc, $wgs workgroups"
    KAI_VICTIM_WGS=$wgs KAI_PROF=1 timeout 900 python bench.py --config C4 --scale $sc --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "kai victim|^\{" | cut -c1-400
  done
done > gpurun_out/${TAG}_c4_multi.txt 2>&1
cat gpurun_out/${TAG}_c4_multi.txt
