#!/bin/bash
# round 6 profiles of config 5 on the library with k_fill_levels / k_plan_gather: (1) per-LAUNCH kernel trace of one cycle grouped into rounds, (2) rocprofv3 --stats summary of the
# default bench command, (3) PMC passes (counters only with --kernel-trace, each set in its own run): HBM traffic and the instruction mix / wave cycles of the fill kernel
TAG=${1:-r06c}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0
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
for r in rows:
    n = name(r)
    if n == "k_plan_setup": flush(); rnd += 1; acc = {}
    if rnd == 0 and not n.startswith(("k_batch", "k_bucket", "k_fill", "k_class")): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(n, [0.0, 0]); a[0] += d; a[1] += 1
flush()
# the plan_scan launches one by one (height 1, 2, 3 of every round)
print("# k_plan_scan launches in order (us):", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in rows if name(r) == "k_plan_scan"))
print("# k_plan_rank launches in order (us):", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in rows if name(r) == "k_plan_rank"))
PY
head -30 gpurun_out/${TAG}_c5_kernels_per_round.txt | cut -c1-400
rm -rf gpurun_out/prof_${TAG}_launches
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_${TAG}_stats" -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-sample 0 > "$R/gpurun_out/${TAG}_stats.log" 2>&1; echo "stats rc=$?"
cd "$R"
s=$(find gpurun_out/prof_${TAG}_stats -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" gpurun_out/${TAG}_c5_kernel_stats.csv && head -12 gpurun_out/${TAG}_c5_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_${TAG}_stats
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  cd /tmp
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_${TAG}_$tag" -- python "$R/bench.py" --config C5 --steps 3 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/pmc_${TAG}_$tag.log" 2>&1; echo "$set rc=$?"
  cd "$R"
  c=$(find gpurun_out/pmc_${TAG}_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$c" ] && cp "$c" gpurun_out/${TAG}_c5_pmc_$tag.csv
  rm -rf gpurun_out/pmc_${TAG}_$tag
done
python - <<PY > gpurun_out/${TAG}_fill_pmc_instruction_mix.txt
import csv, glob, collections
print("# k_fill_levels on config 5 (bench.py --config C5 --steps 3 --warmup 0), rocprofv3 --pmc passes (each counter set in its own run, with --kernel-trace only): sums over the kernel's launches of 3 cycles")
for f in sorted(glob.glob("gpurun_out/${TAG}_c5_pmc_*.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if "k_fill_levels" in row.get("Kernel_Name", ""): acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for k in acc: print(f"{k:24s} {acc[k]:16.0f}  over {n[k]} launches")
PY
cat gpurun_out/${TAG}_fill_pmc_instruction_mix.txt
