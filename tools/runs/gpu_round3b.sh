#!/bin/bash
# round-3 evidence, second set (scan grid): suite, default bench, the mixed config 5 and config 3 with fractions, kernel trace and PMC passes of the fractions run.
# usage: gpu_round3b.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r03x}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as G; G.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.txt | tail -1
KAI_PROF=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/${TAG}_bench_default.json
: > gpurun_out/${TAG}_bench_lines.jsonl
KAI_PROF=1 timeout 900 python bench.py --config C5 --mixed --steps 2 --warmup 1 --cpu-sample 0 2> gpurun_out/${TAG}_mixed.err | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl; grep "scan grid" gpurun_out/${TAG}_mixed.err | tail -1
KAI_SCAN_WGS=1 timeout 900 python bench.py --config C5 --mixed --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl
timeout 900 python bench.py --config C3 --fractions 0.3 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl
KAI_SCAN_WGS=1 timeout 900 python bench.py --config C3 --fractions 0.3 --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl
cut -c1-330 gpurun_out/${TAG}_bench_lines.jsonl
cd /tmp
ARGS="--config C3 --fractions 0.3 --steps 1 --warmup 0 --cpu-sample 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_frac" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/${TAG}_prof_frac.log" 2>&1; echo "trace rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_frac_$ctr" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/${TAG}_prof_frac_$ctr.log" 2>&1; echo "$ctr rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_mixed" -- python "$R/bench.py" --config C5 --mixed --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/${TAG}_prof_mixed.log" 2>&1; echo "mixed trace rc=$?"
find "$R/gpurun_out/${TAG}_prof_frac" -name '*kernel_stats.csv' | head -1 | xargs -r head -4
find "$R/gpurun_out/${TAG}_prof_mixed" -name '*kernel_trace.csv' | head -1 | xargs -r grep -m2 "k_action"
