#!/bin/bash
# per-LAUNCH kernel trace of ONE config-5 cycle (and one config-2 cycle), grouped into rounds
TAG=${1:-r06g}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0
for CFG in ${TRACE_CONFIGS:-C5 C2}; do
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_${TAG}_$CFG" -- python "$R/bench.py" --config $CFG --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/${TAG}_launches_$CFG.log" 2>&1; echo "trace $CFG rc=$?"
cd "$R"
f=$(find gpurun_out/prof_${TAG}_$CFG -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > gpurun_out/${TAG}_${CFG}_kernels_per_round.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].replace("kai::", "").replace("void ", "")
print("# one cycle (bench.py --steps 1 --warmup 0), rocprofv3 --kernel-trace: kernels in launch order, grouped into rounds at every k_plan_setup; microseconds")
rnd, acc, span, prev_end = 0, {}, [None, None], None
def flush():
    global prev_end
    if acc:
        tot = sum(v[0] for v in acc.values())
        print(f"round {rnd:2d}: " + "  ".join(f"{k} {v[0]:.0f} us x{v[1]}" for k, v in acc.items()) + f"  | total {tot:.0f} us, span {(span[1] - span[0]) / 1e3:.0f} us (gaps inside {(span[1] - span[0]) / 1e3 - tot:.0f} us), idle before {((span[0] - prev_end) / 1e3 if prev_end else 0):.0f} us")
        prev_end = span[1]
for r in rows:
    n = name(r)
    if n in ("k_plan_setup", "k_class_capacity") and (n == "k_class_capacity" or "k_class_capacity" not in acc): flush(); rnd += 1; acc = {}; span = [None, None]
    if rnd == 0 and not n.startswith(("k_batch", "k_bucket", "k_fill", "k_class")): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(n, [0.0, 0]); a[0] += d; a[1] += 1
    if span[0] is None: span[0] = int(r["Start_Timestamp"])
    span[1] = int(r["End_Timestamp"])
flush()
for k in ("k_plan_scan", "k_plan_rank", "k_plan_gather", "k_plan_leaf"):
    print(f"# {k} launches in order (us):", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in rows if name(r) == k))
t0 = min(int(r["Start_Timestamp"]) for r in rows if name(r).startswith("k_batch_static_rank")); t1 = max(int(r["End_Timestamp"]) for r in rows if name(r) in ("k_drain", "k_apply_nodes"))
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1)
print(f"# action span {(t1 - t0) / 1e3:.0f} us, kernels busy {busy / 1e3:.0f} us, gaps {(t1 - t0 - busy) / 1e3:.0f} us")
PY
cat gpurun_out/${TAG}_${CFG}_kernels_per_round.txt | cut -c1-420
cp "$f" gpurun_out/${TAG}_${CFG}_kernel_trace.csv 2>/dev/null; rm -rf gpurun_out/prof_${TAG}_$CFG
done
