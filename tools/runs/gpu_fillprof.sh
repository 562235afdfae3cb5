#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in C5 C3; do KAI_CORE_LIB=$PWD/build/libkai_core_prof.so KAI_PROF=1 timeout 600 python bench.py --config $cfg --steps 1 --warmup 1 --cpu-sample 0 2>&1 | grep -E "kai batch" | tail -1; done
bash tools/gpu_pmc_fill.sh 2>&1 | tail -4
