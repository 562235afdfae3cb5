#!/bin/bash
# instruction mix / busy cycles of the victim actions' kernel (k_action<true,…>, 32 workgroups) on BASELINE config 4 at 2 %: rocprofv3 PMC passes, counters only with --kernel-trace
cd /tmp; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; mkdir -p $R/gpurun_out; export KAI_BENCH_OPEN_LEG=0 KAI_BENCH_OTHER_SHAPES=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_victim_$tag" -- python "$R/bench.py" --config C4 --scale 0.02 --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/pmc_victim_$tag.log" 2>&1; echo "$set rc=$?"
done
python - <<'PY'
import csv,glob,os,collections
R=os.environ.get("GRAFT_REPO_ROOT",".")
for f in sorted(glob.glob(R+"/gpurun_out/pmc_victim_*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(float); n=0
    for row in csv.DictReader(open(f)):
        if 'k_action' in row.get('Kernel_Name','') and 'true' in row.get('Kernel_Name','').split('k_action')[1][:12]: acc[row['Counter_Name']]+=float(row['Counter_Value']); n+=1
    print(os.path.basename(os.path.dirname(os.path.dirname(f))), n, dict(acc))
PY
