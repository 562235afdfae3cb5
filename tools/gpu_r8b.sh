#!/bin/bash
# round 6, session 4: shared GPUs with the class index (summary bits in n_flags) and the staged job path for gangs without a fraction pod: config 3 + 30 % fractions at the three levels
# of KAI_SHARED_INDEX (operations hashed against the oracle's pin), the -m gpu tests that hold fractions / GPU memory / MIG, and where the sequential engine's time goes on config 5's
# shape (tools/mixed_variants.py)
TAG=${1:-r08b}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q -k "fraction or memory or mig or shared or smoke or scan_grid" --durations=5 ) > gpurun_out/${TAG}_pytest_fractions.txt 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/${TAG}_pytest_fractions.txt | cut -c1-200
for ix in 2 1 0; do
  KAI_SHARED_INDEX=$ix KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --config C3 --fractions 0.3 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_c3_fractions_index$ix.json 2> gpurun_out/${TAG}_bench_c3_fractions_index$ix.err; echo "c3 fractions index=$ix rc=$?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_c3_fractions_index$ix.json").read().strip().splitlines()[-1])
print("index=$ix ms_per_step", round(d["ms_per_step"], 2), "parity", d.get("parity_full", {}).get("equal_to_oracle"), "ops", d.get("parity_full", {}).get("ops"))
PY
  grep "kai prof" gpurun_out/${TAG}_bench_c3_fractions_index$ix.err | tail -1 | cut -c1-400
done
KAI_PROF=1 timeout 900 python tools/mixed_variants.py > gpurun_out/${TAG}_mixed_variants.txt 2> gpurun_out/${TAG}_mixed_variants.err; echo "variants rc=$?"
cat gpurun_out/${TAG}_mixed_variants.txt; grep "kai prof\|done" gpurun_out/${TAG}_mixed_variants.err | cut -c1-400
