#!/bin/bash
# the whole -m gpu suite + smoke, as the driver runs them at the round's end
TAG=${1:-r06s}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -22 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/${TAG}_smoke.txt
