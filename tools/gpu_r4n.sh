#!/bin/bash
# round 4, final build: the whole GPU suite, the default bench (as the driver runs it), rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the C5 bench, the multi-process rehearsals
TAG=${1:-r04n}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -2 gpurun_out/${TAG}_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.txt
KAI_PROF=1 KAI_BATCH_TRACE=1 timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench default rc=$?"
grep "kai batch (bucket" gpurun_out/${TAG}_bench_default.err | head -3 | tail -1
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_default.json'))
print('C5', round(d['ms_per_step'],2), round(d['value']), d['roofline']['frac'], d['parity_full']['equal_to_oracle'], (d.get('cpu_same_algorithm') or {}).get('ms_per_step'))
print(json.dumps(d.get('cycle_with_open_ms')), json.dumps(d.get('cycle_pipelined_ms')))
for k, v in (d.get('other_shapes') or {}).items(): print(k, json.dumps(v)[:300])
PY
KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 bash tools/runs/gpu_prof.sh ${TAG} 2>&1 | tail -5
python tools/pmc_traffic.py "C5 65536n x 1000000p full chain + time-based fair-share" $(find gpurun_out/prof_${TAG}_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find gpurun_out/prof_${TAG}_WRITE_SIZE -name '*counter_collection.csv' | head -1) k_fill_buckets
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
find gpurun_out/prof_${TAG}_trace -name '*kernel_stats.csv' | head -1 | xargs -r -I{} cp {} gpurun_out/${TAG}_c5_kernel_stats.csv; head -6 gpurun_out/${TAG}_c5_kernel_stats.csv
KAI_BENCH_BACKEND=gloo KAI_BENCH_ONE_DEVICE=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/${TAG}_multiprocess_rehearsal.json 2> gpurun_out/${TAG}_multiprocess_rehearsal.err; echo "rehearsal rc=$?"
tail -c 1200 gpurun_out/${TAG}_multiprocess_rehearsal.json | cut -c1-600
