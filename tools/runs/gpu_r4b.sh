#!/bin/bash
# round 4: the bucket fill (kai_fill_buckets.hpp) on the device — parity of the batch path, the default bench, an A/B against the general kernel, rocprof kernel stats
TAG=${1:-r04b}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bucket or batch or synthetic or full_size or three_level or quarter" > gpurun_out/${TAG}_pytest_batch.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest_batch.txt
KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench rc=$?"; grep "kai batch" gpurun_out/${TAG}_bench_c5.err | tail -1
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_c5.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel'], d['parity_full']['equal_to_oracle'], d.get('cpu_same_algorithm',{}).get('ms_per_step'), json.dumps(d['config']['engine']))
PY
KAI_FILL_GENERAL=1 KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5_general.json 2> gpurun_out/${TAG}_bench_c5_general.err; echo "bench general rc=$?"; grep "kai batch" gpurun_out/${TAG}_bench_c5_general.err | tail -1
for cfg in C2 C3; do KAI_PROF=1 timeout 300 python bench.py --config $cfg --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_${cfg}.json 2> gpurun_out/${TAG}_bench_${cfg}.err; grep "kai batch" gpurun_out/${TAG}_bench_${cfg}.err | tail -1; python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_${cfg}.json')); print('$cfg', d['ms_per_step'], d['value'])"; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_trace" -- python "$R/bench.py" --config C5 --steps 2 --warmup 1 --cpu-sample 0 > "$R/gpurun_out/${TAG}_prof_trace.log" 2>&1; echo "trace rc=$?"
find "$R/gpurun_out/${TAG}_prof_trace" -name '*kernel_stats.csv' | head -1 | xargs -r head -12
