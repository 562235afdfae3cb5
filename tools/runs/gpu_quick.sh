#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/quick_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/quick_pytest_gpu.txt
KAI_PROF=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "kai batch|^\{" | cut -c1-420
