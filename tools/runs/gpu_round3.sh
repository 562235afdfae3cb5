#!/bin/bash
# round-3 evidence set.  usage: gpu_round3.sh <tag>   (writes gpurun_out/<tag>_*)
TAG=${1:-r03x}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as G; G.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.txt | tail -1
KAI_PROF=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/${TAG}_bench_default.json
: > gpurun_out/${TAG}_bench_lines.jsonl
for cfg in C2 C3; do timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl; done
KAI_BENCH_ENGINE_MODE=3 timeout 600 python bench.py --config C5 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl
timeout 900 python bench.py --config C5 --mixed --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl
KAI_PROF=1 timeout 900 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 2> gpurun_out/${TAG}_c4.err | grep '^{' >> gpurun_out/${TAG}_bench_lines.jsonl; grep "kai victim" gpurun_out/${TAG}_c4.err | tail -1
# the multi-process path of bench.py (replicas: one scheduling shard per rank), rehearsed with both ranks on this one GPU over gloo
KAI_BENCH_BACKEND=gloo KAI_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | grep '^{' > gpurun_out/${TAG}_multiprocess_rehearsal.jsonl; echo "rehearsal rc=$?"
cut -c1-260 gpurun_out/${TAG}_bench_lines.jsonl gpurun_out/${TAG}_multiprocess_rehearsal.jsonl
cd /tmp
ARGS="--config C5 --steps 2 --warmup 1 --cpu-sample 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_trace" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/${TAG}_prof_trace.log" 2>&1; echo "trace rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_$ctr" -- python "$R/bench.py" $ARGS > "$R/gpurun_out/${TAG}_prof_$ctr.log" 2>&1; echo "$ctr rc=$?"
done
# the victim action kernel on its 32 workgroups: C4 at 3 %
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_c4" -- python "$R/bench.py" --config C4 --scale 0.03 --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/${TAG}_prof_c4.log" 2>&1; echo "c4 trace rc=$?"
find "$R/gpurun_out/${TAG}_prof_trace" -name '*kernel_stats.csv' | head -1 | xargs -r head -6
find "$R/gpurun_out/${TAG}_prof_c4" -name '*kernel_stats.csv' | head -1 | xargs -r head -4
find "$R/gpurun_out/${TAG}_prof_c4" -name '*kernel_trace.csv' | head -1 | xargs -r grep -m2 "k_action"
