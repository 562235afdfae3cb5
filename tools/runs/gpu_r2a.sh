#!/bin/bash
# round 2, first GPU session: parity of the batch path + first bench lines with its phase breakdown + rocprof kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02e_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02e_pytest_gpu.txt
for cfg in C5 C3 C2; do
  KAI_PROF=1 timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r02e_bench_$cfg.log 2>&1; echo "bench $cfg rc=$?"
  grep -E "kai batch" gpurun_out/r02e_bench_$cfg.log | tail -2; grep '^{' gpurun_out/r02e_bench_$cfg.log | cut -c1-600
done
KAI_BENCH_ENGINE_MODE=3 timeout 600 python bench.py --config C5 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r02e_bench_C5_seq.log 2>&1; grep '^{' gpurun_out/r02e_bench_C5_seq.log | cut -c1-400
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r02e_prof" -- python "$GRAFT_REPO_ROOT/bench.py" --config C5 --steps 2 --warmup 1 --cpu-sample 0 > "$GRAFT_REPO_ROOT/gpurun_out/r02e_prof.log" 2>&1; echo "prof rc=$?"
find "$GRAFT_REPO_ROOT/gpurun_out/r02e_prof" -name '*kernel_stats.csv' | head -1 | xargs -r head -30
