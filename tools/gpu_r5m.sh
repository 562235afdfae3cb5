#!/bin/bash
TAG=${1:-r05m}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "victim_waves_over_the_ranks" > gpurun_out/${TAG}_ranks_$i.txt 2>&1; echo "no -s run $i rc=$?"; grep -n "rank .* failed\|passed\|failed" gpurun_out/${TAG}_ranks_$i.txt | head -4 | cut -c1-300; done
for i in 4 5; do timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "victim_waves_over_the_ranks" > gpurun_out/${TAG}_ranks_$i.txt 2>&1; echo "-s run $i rc=$?"; grep -n "rank .* failed\|passed\|failed" gpurun_out/${TAG}_ranks_$i.txt | head -4 | cut -c1-300; done
