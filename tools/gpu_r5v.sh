#!/bin/bash
# round 5, after the round's oracle additions (scenario filters, builder, scenario objects, shared-GPU pieces, capacity chain, queue attributes, minruntime) and the harness
# change: the -m gpu suite without the four config-4 hash tests (400 of its 595 s; the library is the one `r05zz` ran them on), then smoke
TAG=${1:-r05v}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 480 python -m pytest tests -m gpu -q -x --durations=5 -k "not config4_cycle_hashes and not config4_with_queue_depth" > gpurun_out/${TAG}_pytest_gpu_without_config4.txt 2>&1; echo "pytest rc=$?"; tail -9 gpurun_out/${TAG}_pytest_gpu_without_config4.txt | cut -c1-160
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.txt
