#!/bin/bash
# instruction mix of the fill kernel (rocprofv3 PMC passes, counters only with --kernel-trace)
cd /tmp; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; mkdir -p $R/gpurun_out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fill_$tag" -- python "$R/bench.py" --config C5 --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/pmc_fill_$tag.log" 2>&1; echo "$set rc=$?"
done
python - <<'PY'
import csv,glob,os,collections
R=os.environ.get("GRAFT_REPO_ROOT",".")
for f in sorted(glob.glob(R+"/gpurun_out/pmc_fill_*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if 'k_fill' in row.get('Kernel_Name',''): acc[row['Counter_Name']]+=float(row['Counter_Value'])
    print(os.path.basename(os.path.dirname(os.path.dirname(f))), dict(acc))
PY
