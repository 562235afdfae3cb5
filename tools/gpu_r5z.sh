#!/bin/bash
# round 5, closing set on the round's last build: the whole -m gpu suite, smoke, then tools/gpu_r5f.sh (rocprofv3 kernel stats, FETCH / WRITE passes, instruction mix of the fill kernel, the default bench line with the driver's flags)
TAG=${1:-r05z}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -12 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-160
bash tools/gpu_r5f.sh ${TAG}
