#!/bin/bash
# round 6, first device pass of the library with k_fill_levels (kai_fill_levels.hpp): the fill tests of the -m gpu suite, the default bench line, config 2, the rounds of one config-5 cycle
TAG=${1:-r06a}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config1 or config2 or config5 or counts or bucket or decisions_close or random_small" > gpurun_out/${TAG}_pytest_subset.txt 2>&1; echo "pytest subset rc=$?"; tail -4 gpurun_out/${TAG}_pytest_subset.txt | cut -c1-160
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"
timeout 120 python bench.py --config C2 --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "bench c2 rc=$?"
KAI_BATCH_TRACE=1 timeout 120 python bench.py --config C5 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> gpurun_out/${TAG}_c5_rounds.txt; echo "rounds rc=$?"
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_default.json", "gpurun_out/${TAG}_bench_c2.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        e = d["config"]["engine"]
        print(f, "ms_per_step", round(d["ms_per_step"], 3), "value", round(d["value"]), "kernel", e.get("fill_kernel"), "plan/fill/apply ms", e.get("plan_ms"), e.get("fill_ms"), e.get("apply_ms"), "rounds", e.get("rounds"), "parity", d.get("parity_full", {}).get("equal_to_oracle"), "open p50", d.get("cycle_with_open_ms", {}).get("p50"))
        o = d.get("other_shapes", {})
        for k, v in o.items():
            if isinstance(v, dict) and "ms_per_step" in v: print("   ", k, round(v["ms_per_step"], 2), v.get("equal_to_oracle_pin", v.get("ops_equal_to_pin")))
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
grep "kai batch round" gpurun_out/${TAG}_c5_rounds.txt | head -20
