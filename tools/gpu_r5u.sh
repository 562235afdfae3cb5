#!/bin/bash
# round 5, measurements that need no rebuild: (1) the rounds' first plan depth (KAI_BATCH_H0, default 256) swept on config 5; (2) a per-LAUNCH kernel trace of ONE config-5
# cycle, so that the plan / fill / apply kernels' time can be read per round (profiles/ holds the per-kernel sums only)
TAG=${1:-r05u}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0
for h in 64 256 1024 4096; do
  KAI_BATCH_H0=$h timeout 120 python bench.py --config C5 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5_h0_$h.json 2> gpurun_out/${TAG}_bench_c5_h0_$h.err; echo "H0=$h rc=$?"
done
python - <<PY
import json
for h in (64, 256, 1024, 4096):
    d = json.loads(open(f"gpurun_out/${TAG}_bench_c5_h0_{h}.json").read().strip().splitlines()[-1]); e = d["config"]["engine"]
    print("H0", h, "ms_per_step", round(d["ms_per_step"], 2), "rounds", e["rounds"], "plan", e["plan_ms"], "fill", e["fill_ms"], "apply", e["apply_ms"], "mispredicted", e["mispredicted_jobs"], d["parity_full"]["equal_to_oracle"] if "parity_full" in d else None)
PY
cd /tmp
KAI_BATCH_TRACE=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_${TAG}_launches" -- python "$R/bench.py" --config C5 --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/${TAG}_launches.log" 2> "$R/gpurun_out/${TAG}_launches.err"; echo "trace rc=$?"
cd "$R"
f=$(find gpurun_out/prof_${TAG}_launches -name '*kernel_trace.csv' | head -1); echo "trace file: $f"
python - "$f" <<'PY' > gpurun_out/${TAG}_c5_kernels_per_round.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].replace("kai::", "").replace("void ", "")
print("# one config-5 cycle (bench.py --config C5 --steps 1 --warmup 0), rocprofv3 --kernel-trace: kernels in launch order, grouped into rounds at every k_plan_setup; microseconds")
rnd, acc = 0, {}
def flush():
    if acc: print(f"round {rnd:2d}: " + "  ".join(f"{k} {v[0]:.0f} us x{v[1]}" for k, v in acc.items()) + f"  | total {sum(v[0] for v in acc.values()):.0f} us")
seen_action = False
for r in rows:
    n = name(r)
    if n == "k_plan_setup": flush(); rnd += 1; acc = {}
    if rnd == 0 and not n.startswith(("k_batch", "k_bucket", "k_fill", "k_class")): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(n, [0.0, 0]); a[0] += d; a[1] += 1
flush()
PY
head -30 gpurun_out/${TAG}_c5_kernels_per_round.txt | cut -c1-330
grep "kai batch round" gpurun_out/${TAG}_launches.err | head -14 > gpurun_out/${TAG}_c5_rounds.txt
rm -rf gpurun_out/prof_${TAG}_launches
