#!/bin/bash
# round-3: topology / victim tests + C4 10 % + mixed C5 after the domain loops went to the scan lanes and the early reject.  usage: gpu_r3i.sh <tag>
TAG=${1:-r03i}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "opolog or several_workgroups or golden or integration or campaign or broad or synthetic or victim or ingested" > gpurun_out/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_pytest.txt | tail -1
: > gpurun_out/${TAG}_bench_lines.jsonl
KAI_PROF=1 timeout 900 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 --cpu-sample 0 2> gpurun_out/${TAG}_c4.err | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl; grep "kai victim" gpurun_out/${TAG}_c4.err | tail -1
timeout 900 python bench.py --config C4 --scale 0.03 --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl
timeout 900 python bench.py --config C5 --mixed --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl
cut -c1-240 gpurun_out/${TAG}_bench_lines.jsonl
