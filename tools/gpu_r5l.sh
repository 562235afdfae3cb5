#!/bin/bash
TAG=${1:-r05l}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "victim_waves_over_the_ranks" > gpurun_out/${TAG}_ranks_$i.txt 2>&1; echo "run $i rc=$?"; tail -2 gpurun_out/${TAG}_ranks_$i.txt | cut -c1-200; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "(victim or golden or reclaim_large or memory_flat or default_cycle) and not victim_waves_over_the_ranks" > gpurun_out/${TAG}_pytest_victim.txt 2>&1; echo "pytest victim rc=$?"; tail -2 gpurun_out/${TAG}_pytest_victim.txt
