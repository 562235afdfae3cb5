#!/bin/bash
# instruction mix and cache behaviour of the victim-action kernel (k_action, reclaim of BASELINE config 4 at 3 %): rocprofv3 PMC passes, counters only with --kernel-trace
cd /tmp; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; mkdir -p $R/gpurun_out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_victim_$tag" -- python "$R/bench.py" --config C4 --scale 0.03 --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/pmc_victim_$tag.log" 2>&1; echo "$set rc=$?"
done
python - <<'PY' | tee $R/gpurun_out/pmc_victim_summary.txt
import csv,glob,os,collections
R=os.environ.get("GRAFT_REPO_ROOT",".")
print("# rocprofv3 --pmc passes of: bench.py --config C4 --scale 0.03 --steps 1 --warmup 0 (allocate, consolidation, reclaim); per kernel name, summed over its launches and XCDs")
for f in sorted(glob.glob(R+"/gpurun_out/pmc_victim_*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        k=row.get('Kernel_Name','')
        if 'k_action' in k or 'k_victim' in k: acc[k.split('(')[0]][row['Counter_Name']]+=float(row['Counter_Value'])
    for k,v in acc.items(): print(k, dict(v))
PY
