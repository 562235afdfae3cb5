#!/bin/bash
# round-2 evidence set.  usage: gpu_round2.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r02x}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as G; G.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${TAG}_pytest_gpu.txt
KAI_PROF=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/${TAG}_bench_default.json
: > gpurun_out/${TAG}_bench_lines.jsonl
for cfg in C2 C3; do timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl; done
KAI_BENCH_ENGINE_MODE=3 timeout 600 python bench.py --config C5 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl
cut -c1-300 gpurun_out/${TAG}_bench_lines.jsonl
cd /tmp
ARGS="--config C5 --steps 2 --warmup 1 --cpu-sample 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_trace" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/${TAG}_prof_trace.log" 2>&1; echo "trace rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_$ctr" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/${TAG}_prof_$ctr.log" 2>&1; echo "$ctr rc=$?"
done
find "$R/gpurun_out/${TAG}_prof_trace" -name '*kernel_stats.csv' | head -1 | xargs -r head -8
