#!/bin/bash
# round 6, session 4: the library with the class index kept in shared-GPU sessions — the whole -m gpu suite + smoke as the driver runs them, the default bench line (other_shapes carries
# c3_fractions_30), device campaigns with fractions / GPU memory / MIG and the broad cases against the oracle
TAG=${1:-r08c}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -22 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/${TAG}_smoke.txt
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print("C5 ms_per_step", round(d["ms_per_step"], 2), "value", round(d["value"]), "scaling", d["scaling"], "parity", d["parity_full"]["equal_to_oracle"], "roofline", d["roofline"]["bound"], round(d["roofline"]["frac"], 3))
for k, v in d.get("other_shapes", {}).items():
    if "ms_per_step" in v: print(" ", k, round(v["ms_per_step"], 2), "ms", v.get("path"), "equal_to_oracle", v.get("equal_to_oracle"))
for k, v in d.get("other_shapes", {}).get("reference_benchmarks", {}).items(): print(" ", k, round(v["open_plus_actions_ms"], 1), "ms equal_to_oracle", v["equal_to_oracle"])
PY
CAMPAIGN_SECONDS=150 CAMPAIGN_SECONDS_MIG=240 SEED_BROAD=930000 SEED_MIG=41000 bash tools/gpu_final_campaign.sh ${TAG} 2>&1 | tail -8
