#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; T=${1:-r03l}
bash tools/gpu_tprof.sh 1.0 > gpurun_out/${T}_tprof.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mixed_shape or config4 or replica_groups or elastic_and_subgroups or broad_random or reference_goldens" > gpurun_out/${T}_pytest_topo.txt 2>&1
tail -3 gpurun_out/${T}_pytest_topo.txt
timeout 300 python bench.py --config C5 --mixed --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${T}_bench_mixed.json 2>&1
cut -c1-400 gpurun_out/${T}_bench_mixed.json
grep "kai prof" gpurun_out/${T}_tprof.txt
timeout 600 python bench.py --config C3 --fractions 0.3 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${T}_bench_c3_fractions.json 2>&1
cut -c1-300 gpurun_out/${T}_bench_c3_fractions.json; grep -o '"scanner".*' gpurun_out/${T}_bench_c3_fractions.json | cut -c1-400
