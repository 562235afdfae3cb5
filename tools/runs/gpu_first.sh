#!/bin/bash
# first GPU pass: parity suite, bench lines, kernel trace.  Every leg has its own timeout.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rocminfo | grep -m1 gfx > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt; grep -m1 'model name' /proc/cpuinfo >> gpurun_out/gpu.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --config C2 --steps 3 --warmup 1 > gpurun_out/bench_c2.log 2>&1; echo "bench c2 rc=$?"; tail -2 gpurun_out/bench_c2.log
timeout 600 python bench.py --config C5 --scale 0.05 --steps 1 --warmup 0 > gpurun_out/bench_c5s.log 2>&1; echo "bench c5 x0.05 rc=$?"; tail -2 gpurun_out/bench_c5s.log
R="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_c2" -- python "$R/bench.py" --config C2 --steps 2 --warmup 1 --cpu-sample 0 > "$R/gpurun_out/prof_c2.log" 2>&1; echo "rocprof rc=$?"
find "$R/gpurun_out/prof_c2" -name '*stats*' | head
