#!/bin/bash
# round 5, last session (written WITHOUT a GPU: to be run first thing when one is available): what the rewritten kai_session_open costs on the box.
#   1. the GPU tests that open the largest sessions (the snapshot's own arrays go up from pinned staging from 4 MB on) + smoke
#   2. the default bench line (cycle_with_open_ms.open_p50 is the figure) with KAI_PROF's host clocks of every open on stderr
#   3. the same with KAI_OPEN_FULL_UPLOADS=1 (every array sent), KAI_OPEN_NO_STAGING=1 (no pinned staging), KAI_HOST_POOL=0 (threads started per loop): what each change buys
TAG=${1:-r05open}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as G; G.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size or synthetic_configs or memory_flat or config2 or config5" > gpurun_out/${TAG}_pytest_subset.txt 2>&1; echo "pytest subset rc=$?"; tail -3 gpurun_out/${TAG}_pytest_subset.txt | cut -c1-160
run() { name=$1; shift; env "$@" KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_${name}.json 2> gpurun_out/${TAG}_bench_${name}.err; echo "bench $name rc=$?"; grep 'kai open' gpurun_out/${TAG}_bench_${name}.err | tail -4 | cut -c1-260; }
run default KAI_X=0
run full_uploads KAI_OPEN_FULL_UPLOADS=1
run no_staging KAI_OPEN_NO_STAGING=1
run no_pool KAI_HOST_POOL=0
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default_all_shapes.json 2> gpurun_out/${TAG}_bench_default_all_shapes.err; echo "bench (driver flags) rc=$?"
python - <<PY
import json
for n in ("default", "full_uploads", "no_staging", "no_pool", "default_all_shapes"):
    try:
        d = json.loads(open("gpurun_out/${TAG}_bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms_per_step", round(d["ms_per_step"], 2), "open_p50", d.get("cycle_with_open_ms", {}).get("open_p50"), "cycle_with_open", d.get("cycle_with_open_ms", {}).get("p50"), "pipelined", d.get("cycle_pipelined_ms", {}).get("p50"),
              "equal_to_oracle", d.get("parity_full", {}).get("equal_to_oracle"), {k: (v.get("open_plus_actions_ms"), v.get("equal_to_oracle")) for k, v in (d.get("other_shapes", {}).get("reference_benchmarks") or {}).items() if isinstance(v, dict)})
    except Exception as e:
        print(n, "unreadable:", e)
PY
