#!/bin/bash
# round 5, the library with the adopted round policy (full depth at once after a round without a surprise, a quarter after a plan mostly thrown away): the -m gpu suite
# without the four config-4 hash tests (400 of its 595 s; the victim search is untouched), then the bench line
TAG=${1:-r05r}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 230 python -m pytest tests -m gpu -q -x --durations=3 -k "not config4_cycle_hashes and not config4_with_queue_depth" > gpurun_out/${TAG}_pytest_gpu_without_config4.txt 2>&1; echo "pytest rc=$?"; tail -7 gpurun_out/${TAG}_pytest_gpu_without_config4.txt | cut -c1-160
KAI_BENCH_OTHER_SHAPES=0 timeout 60 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1]); e = d["config"]["engine"]
print(d["ms_per_step"], d["value"], e["rounds"], e["plan_ms"], e["fill_ms"], d["parity_full"]["equal_to_oracle"], (d.get("cycle_with_open_ms") or {}).get("p50"), (d.get("cycle_pipelined_ms") or {}).get("p50"))
PY
