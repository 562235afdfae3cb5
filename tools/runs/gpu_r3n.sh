#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; T=${1:-r03n}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scan_grid" > gpurun_out/${T}_pytest_grid.txt 2>&1
tail -5 gpurun_out/${T}_pytest_grid.txt
bash tools/gpu_tprof.sh 1.0 > gpurun_out/${T}_tprof.txt 2>&1
grep "kai prof\|scan grid" gpurun_out/${T}_tprof.txt
for w in 1 8 16 32 64 128; do
  echo "== scan wgs $w"
  KAI_SCAN_WGS=$w timeout 300 python bench.py --config C5 --mixed --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -o '"ms_per_step": [0-9.]*' 
  KAI_SCAN_WGS=$w timeout 600 python bench.py --config C3 --fractions 0.3 --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
