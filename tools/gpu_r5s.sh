#!/bin/bash
# round 5, A/B run: how the plan depth H moves between rounds (kai_batch_driver.hpp: x2 after a round without a surprise, /2 after a plan mostly thrown away), on a build
# that reads KAI_BATCH_POLICY (KAI_CORE_LIB): 0 = as in the library, 1 = x4 up, 2 = back to 256 at once, 3 = x4 up and /4 down
TAG=${1:-r05s}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 KAI_CORE_LIB=$R/kai-scheduler_amd/csrc/libkai_core_ab.so
for cfgs in "C5 10" "C3 10" "C2 30"; do set -- $cfgs
for p in 2 3 4 2 4; do
  KAI_BATCH_POLICY=$p timeout 120 python bench.py --config $1 --steps $2 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_$1_policy_$p.json 2> gpurun_out/${TAG}_$1_policy_$p.err; rc=$?
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_$1_policy_$p.json").read().strip().splitlines()[-1]); e = d["config"]["engine"]
print("$1 policy", $p, "rc", $rc, "ms_per_step", round(d["ms_per_step"], 3), "plan", e.get("plan_ms"), "fill", e.get("fill_ms"), "rounds", e.get("rounds"), (d.get("parity_full") or {}).get("equal_to_oracle"))
PY
done; done
