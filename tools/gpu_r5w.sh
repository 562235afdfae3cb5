#!/bin/bash
# round 5, after the harness fix (bench.py keeps a view of the C ABI's operation array instead of a structured numpy copy, 3.5 ms per config-5 step): the tests that read
# operations through Session.execute on both paths, then the default bench line twice (once with the host clocks of every step on stderr)
TAG=${1:-r05w}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config1 or config2 or config5 or counts or decisions_close or random_small" > gpurun_out/${TAG}_pytest_subset.txt 2>&1; echo "pytest subset rc=$?"; tail -4 gpurun_out/${TAG}_pytest_subset.txt | cut -c1-160
KAI_BENCH_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default_traced.json 2> gpurun_out/${TAG}_bench_default_traced.err; echo "bench traced rc=$?"
tail -6 gpurun_out/${TAG}_bench_default_traced.err | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_default_traced.json", "gpurun_out/${TAG}_bench_default.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], d["config"]["action_ms"], d.get("parity_full", {}).get("equal"), d.get("cycle_with_open_ms", {}).get("p50"), d.get("cycle_pipelined_ms", {}).get("p50"))
PY
