#!/bin/bash
# round 5, A/B run: the workgroup of k_plan_scan (KB_PLAN_SCAN_THREADS = 512) overridden through KAI_PLAN_SCAN_THREADS on a build that reads the variable (KAI_CORE_LIB)
TAG=${1:-r05t}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 KAI_CORE_LIB=$R/kai-scheduler_amd/csrc/libkai_core_ab.so
for t in 512 1024 768 256 1024 512; do
  KAI_PLAN_SCAN_THREADS=$t timeout 120 python bench.py --config C5 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_c5_scan_$t.json 2> gpurun_out/${TAG}_c5_scan_$t.err; rc=$?
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_c5_scan_$t.json").read().strip().splitlines()[-1]); e = d["config"]["engine"]
print("threads", $t, "rc", $rc, "ms_per_step", round(d["ms_per_step"], 2), "plan", e["plan_ms"], "fill", e["fill_ms"], "rounds", e["rounds"], d["parity_full"]["equal_to_oracle"])
PY
done
