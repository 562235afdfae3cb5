#!/bin/bash
# round-3 closing evidence on the final build: smoke, suite, default bench (with other_shapes), rocprof kernel stats of the default bench, a device campaign.
# usage: gpu_round3c.sh <tag>
TAG=${1:-r03z}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as G; G.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.txt | tail -1
KAI_PROF=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench_default.json
python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_default.json')); print(json.dumps(d.get('other_shapes'))[:900]); print(d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_full']['equal_to_oracle'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_trace" -- python "$R/bench.py" --config C5 --steps 2 --warmup 1 --cpu-sample 0 > "$R/gpurun_out/${TAG}_prof_trace.log" 2>&1; echo "trace rc=$?"
find "$R/gpurun_out/${TAG}_prof_trace" -name '*kernel_stats.csv' | head -1 | xargs -r head -6
cd "$R"
KAI_SCAN_WGS=8 SEED_BROAD=500000 SEED_MIG=60000 CAMPAIGN_SECONDS=100 CAMPAIGN_SECONDS_MIG=60 bash tools/gpu_final_campaign.sh ${TAG} | tail -8
