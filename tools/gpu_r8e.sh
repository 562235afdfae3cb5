#!/bin/bash
# round 6, session 4, closing set of the final library (follow step published only when the next task is not expected to follow; caller's node flags masked): the shapes the
# sequential engine serves hashed against their pins, host clocks of the small victim-action benchmarks, then the whole -m gpu suite + smoke + the default bench line + device campaigns
TAG=${1:-r08f}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --config C3 --fractions 0.3 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_c3_fractions.json 2> gpurun_out/${TAG}_bench_c3_fractions.err; echo "c3 fractions rc=$?"
KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --config C5 --mixed --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5_mixed.json 2> gpurun_out/${TAG}_bench_c5_mixed.err; echo "c5 mixed rc=$?"
python - <<PY
import json
for f in ("c3_fractions", "c5_mixed"):
    d = json.loads(open(f"gpurun_out/${TAG}_bench_{f}.json").read().strip().splitlines()[-1])
    print(f, "ms_per_step", round(d["ms_per_step"], 2), "parity", d.get("parity_full", {}).get("equal_to_oracle"))
PY
grep "kai prof" gpurun_out/${TAG}_bench_c3_fractions.err | tail -1 | cut -c1-300; grep "kai prof" gpurun_out/${TAG}_bench_c5_mixed.err | tail -1 | cut -c1-300
for b in PreemptAction ConsolidationAction FullSchedulingCycle; do KAI_PROF=1 timeout 120 python tools/prof_reclaim.py $b 2>&1 | grep -v "kai open: host prep" | tail -6; done > gpurun_out/${TAG}_small_benchmarks_host_clocks.txt 2>&1; cut -c1-260 gpurun_out/${TAG}_small_benchmarks_host_clocks.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -16 gpurun_out/${TAG}_pytest_gpu.txt | cut -c1-200
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
cd /tmp; KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_${TAG}_stats" -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sample 0 > "$R/gpurun_out/${TAG}_stats.log" 2>&1; echo "stats rc=$?"; cd "$R"
s=$(find gpurun_out/prof_${TAG}_stats -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" gpurun_out/${TAG}_c5_kernel_stats.csv && head -8 gpurun_out/${TAG}_c5_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_${TAG}_stats
CAMPAIGN_SECONDS=150 CAMPAIGN_SECONDS_MIG=200 SEED_BROAD=960000 SEED_MIG=61000 bash tools/gpu_final_campaign.sh ${TAG} 2>&1 | tail -8
