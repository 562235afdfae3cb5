#!/bin/bash
# round-3 pass: full GPU suite + default bench (C5) + C2 / C3 lines + C4 at 10 %.  usage: gpu_r3f.sh <tag>
TAG=${1:-r03f}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as G; G.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest_gpu.txt | head -2; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.txt | tail -1
KAI_PROF=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-1200 gpurun_out/${TAG}_bench_default.json; grep "kai batch" gpurun_out/${TAG}_bench_default.err | tail -1
: > gpurun_out/${TAG}_bench_lines.jsonl
for cfg in C2 C3; do timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl; done
KAI_PROF=1 timeout 900 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 --cpu-sample 0 2> gpurun_out/${TAG}_c4.err | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl; grep "kai victim" gpurun_out/${TAG}_c4.err | tail -1
cut -c1-260 gpurun_out/${TAG}_bench_lines.jsonl
