#!/bin/bash
# round 5: the victims queue filled best first + index loops from 48 entries on the scan lanes: the reference's small victim benchmarks, the victim-action tests
TAG=${1:-r05k}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/ref_benchmarks.py --max-nodes 500 --iters 3 --out gpurun_out/${TAG}_reference_benchmarks.json > gpurun_out/${TAG}_reference_benchmarks.log 2>&1; echo "ref benchmarks rc=$?"
grep -o '"benchmark": "[A-Za-z_0-9]*"\|"mi355x_open_plus_actions_ms": [0-9.]*' gpurun_out/${TAG}_reference_benchmarks.log | paste - - | tail -18
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "victim or golden or reclaim_large or memory_flat or default_cycle or (config4_cycle_hashes and not 100pct)" > gpurun_out/${TAG}_pytest_victim.txt 2>&1; echo "pytest victim rc=$?"; tail -2 gpurun_out/${TAG}_pytest_victim.txt
