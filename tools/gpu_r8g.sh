#!/bin/bash
# round 6, session 4: PMC passes of the sequential engine's kernel on config 3 + 30 % fractions with the class index kept (each counter set in its own rocprofv3 run, with
# --kernel-trace only): HBM traffic per launch (folded into profiles/pmc_traffic.json = bench.py's roofline.traffic for this workload) and how busy the resident wavefronts are
TAG=${1:-r08g}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  cd /tmp
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_${TAG}_$tag" -- python "$R/bench.py" --config C3 --fractions 0.3 --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/pmc_${TAG}_$tag.log" 2>&1; echo "$set rc=$?"
  cd "$R"
  c=$(find gpurun_out/pmc_${TAG}_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$c" ] && cp "$c" gpurun_out/${TAG}_c3_fractions_pmc_$tag.csv
  rm -rf gpurun_out/pmc_${TAG}_$tag
done
python - <<PY > gpurun_out/${TAG}_c3_fractions_pmc_summary.txt
import csv, glob, collections
print("# k_action<allocate> on config 3 + 30 % fractions with the class index kept (bench.py --config C3 --fractions 0.3 --steps 1 --warmup 0), rocprofv3 --pmc passes, each counter set in its own run: sums over the kernel's launches")
for f in sorted(glob.glob("gpurun_out/${TAG}_c3_fractions_pmc_*.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if "k_action" in row.get("Kernel_Name", ""): acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for k in acc: print(f"{k:24s} {acc[k]:18.0f}  over {n[k]} launches")
PY
cat gpurun_out/${TAG}_c3_fractions_pmc_summary.txt
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic_before.json
python tools/pmc_traffic.py "C3 10000n x 100000p 2-level queues + DRF + 30 % of the one-GPU pods as fractions of one device" gpurun_out/${TAG}_c3_fractions_pmc_FETCH_SIZE.csv gpurun_out/${TAG}_c3_fractions_pmc_WRITE_SIZE.csv k_action && cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
