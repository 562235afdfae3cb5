#!/bin/bash
# round 6: the round loop on the device against the loop on the host (KAI_BATCH_HOST_LOOP=1), same library: fill / batch tests of the -m gpu suite, config 5 and config 2 bench lines in both modes,
# the rounds of one config-5 cycle; $2 = an extra environment assignment for a second A/B (e.g. KAI_PLAN_SEG_MIN=1000000000)
TAG=${1:-r06u}; AB=${2:-}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config1 or config2 or config3 or config5 or counts or bucket or decisions_close or random_small or batch or round_loop or segments" > gpurun_out/${TAG}_pytest_subset.txt 2>&1; echo "pytest subset rc=$?"; tail -4 gpurun_out/${TAG}_pytest_subset.txt | cut -c1-160
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"
timeout 120 python bench.py --config C2 --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "bench c2 rc=$?"
KAI_BENCH_OTHER_SHAPES=0 KAI_BATCH_HOST_LOOP=1 timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${TAG}_hostloop_bench_default.json 2> gpurun_out/${TAG}_hostloop_bench_default.err; echo "bench (loop on the host) rc=$?"
KAI_BENCH_OTHER_SHAPES=0 KAI_BATCH_HOST_LOOP=1 timeout 120 python bench.py --config C2 --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${TAG}_hostloop_bench_c2.json 2> gpurun_out/${TAG}_hostloop_bench_c2.err; echo "bench c2 (loop on the host) rc=$?"
if [ -n "$AB" ]; then
  env KAI_BENCH_OTHER_SHAPES=0 $AB timeout 200 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${TAG}_ab_bench_default.json 2> gpurun_out/${TAG}_ab_bench_default.err; echo "bench ($AB) rc=$?"
  env KAI_BENCH_OTHER_SHAPES=0 $AB timeout 120 python bench.py --config C2 --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${TAG}_ab_bench_c2.json 2> gpurun_out/${TAG}_ab_bench_c2.err; echo "bench c2 ($AB) rc=$?"
fi
KAI_BENCH_OTHER_SHAPES=0 KAI_BATCH_TRACE=1 timeout 120 python bench.py --config C5 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> gpurun_out/${TAG}_c5_rounds.txt; echo "rounds rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        e = d["config"]["engine"]
        print(f, "ms_per_step", round(d["ms_per_step"], 3), "value", round(d["value"]), "kernel", e.get("fill_kernel"), "loop", (e.get("round_loop") or "?")[:13], "plan/fill/apply ms", e.get("plan_ms"), e.get("fill_ms"), e.get("apply_ms"), "rounds", e.get("rounds"), "parity", d.get("parity_full", {}).get("equal_to_oracle"), "open p50", d.get("cycle_with_open_ms", {}).get("p50"))
        o = d.get("other_shapes", {})
        for k, v in o.items():
            if isinstance(v, dict) and "ms_per_step" in v: print("   ", k, round(v["ms_per_step"], 2), v.get("equal_to_oracle"))
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
grep "kai batch round" gpurun_out/${TAG}_c5_rounds.txt | head -12
