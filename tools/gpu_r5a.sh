#!/bin/bash
# round 5, first device pass: the victims log on the MI355X — the reference's own benchmark shapes, config 4 at 10 %, the victim-search tests
TAG=${1:-r05a}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/ref_benchmarks.py --max-nodes 1000 --iters 2 --out gpurun_out/${TAG}_reference_benchmarks.json > gpurun_out/${TAG}_reference_benchmarks.log 2>&1; echo "ref benchmarks rc=$?"
grep -o '"benchmark": "[A-Za-z_0-9]*"\|"mi355x_open_plus_actions_ms": [0-9.]*\|"host_compiled_engine_ms": [0-9.]*' gpurun_out/${TAG}_reference_benchmarks.log | paste - - - | tail -20
KAI_PROF=1 timeout 600 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_10pct.json 2> gpurun_out/${TAG}_c4_10pct.err; echo "c4 10% rc=$?"
grep "kai victim\|kai action host" gpurun_out/${TAG}_c4_10pct.err | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "victim or golden or config4_cycle_hashes_to_the_oracles and not 100pct" > gpurun_out/${TAG}_pytest_victim.txt 2>&1; echo "pytest victim rc=$?"; tail -3 gpurun_out/${TAG}_pytest_victim.txt
