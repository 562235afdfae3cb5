#!/bin/bash
# round 5, final build: rocprofv3 kernel trace + stats of the C5 bench, FETCH_SIZE / WRITE_SIZE passes, instruction mix of k_fill_counts, the default bench line as the driver runs it, smoke
TAG=${1:-r05f}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0
bash tools/runs/gpu_prof.sh ${TAG} --config C5 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -4
python tools/pmc_traffic.py "C5 65536n x 1000000p full chain + time-based fair-share" $(find gpurun_out/prof_${TAG}_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find gpurun_out/prof_${TAG}_WRITE_SIZE -name '*counter_collection.csv' | head -1) k_fill_counts
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
find gpurun_out/prof_${TAG}_trace -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} gpurun_out/${TAG}_c5_kernel_stats.csv; head -8 gpurun_out/${TAG}_c5_kernel_stats.csv
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_${TAG}_$tag" -- python "$R/bench.py" --config C5 --steps 1 --warmup 0 --cpu-sample 0 > "$R/gpurun_out/pmc_${TAG}_$tag.log" 2>&1; echo "$set rc=$?"
done
cd "$R"
python - <<PY > gpurun_out/${TAG}_fill_pmc_instruction_mix.txt
import csv,glob,os,collections
print("# rocprofv3 --pmc <set> --kernel-trace -- python bench.py --config C5 --steps 1 --warmup 0 --cpu-sample 0 (tools/gpu_r5f.sh; three passes; counters of the k_fill_counts launches of ONE C5 cycle, raw values summed over the launches)")
for f in sorted(glob.glob("gpurun_out/pmc_${TAG}_*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(float); n=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if 'k_fill_counts' in row.get('Kernel_Name',''): acc[row['Counter_Name']]+=float(row['Counter_Value']); n[row['Counter_Name']]+=1
    print({k:(v,n[k]) for k,v in acc.items()})
PY
cat gpurun_out/${TAG}_fill_pmc_instruction_mix.txt
unset KAI_BENCH_OTHER_SHAPES KAI_BENCH_OPEN_LEG KAI_BENCH_NATIVE_FILL
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench default rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_default.json'))
print('C5', round(d['ms_per_step'],2), round(d['value']), d['roofline']['frac'], d['parity_full']['equal_to_oracle'])
print(json.dumps(d.get('cycle_with_open_ms')), json.dumps(d.get('cycle_pipelined_ms')))
print(json.dumps(d.get('cpu_sequential_engine'))[:300]); print(json.dumps(d.get('cpu_same_algorithm'))[:200])
for k, v in (d.get('other_shapes') or {}).items(): print(k, json.dumps(v)[:300])
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.txt
