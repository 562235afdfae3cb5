#!/bin/bash
# round 4: fill v3 + host clocks of open / action + PMC passes of k_fill_buckets + rehearsal of the default multi-process (node-sharded) bench on one device over gloo
TAG=${1:-r04d}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bucket or batch or synthetic or full_size or three_level or quarter or node_sharded" > gpurun_out/${TAG}_pytest_batch.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest_batch.txt
KAI_PROF=1 KAI_BATCH_TRACE=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench rc=$?"; grep "kai batch" gpurun_out/${TAG}_bench_c5.err | grep -v round | tail -1
grep -E "kai open|kai action host" gpurun_out/${TAG}_bench_c5.err | tail -4
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_c5.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernel'], d['parity_full']['equal_to_oracle'], d.get('cpu_same_algorithm',{}).get('ms_per_step'))
print(json.dumps(d.get('cycle_with_open_ms')), json.dumps(d.get('cycle_pipelined_ms')))
PY
for cfg in C2 C3; do KAI_PROF=1 timeout 300 python bench.py --config $cfg --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_${cfg}.json 2> gpurun_out/${TAG}_bench_${cfg}.err; python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_${cfg}.json')); print('$cfg', d['ms_per_step'], d['value'], d.get('cycle_with_open_ms'))"; done
KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 bash tools/gpu_prof.sh ${TAG} 2>&1 | tail -6
python tools/pmc_traffic.py "C5 65536n x 1000000p full chain + time-based fair-share" $(find gpurun_out/prof_${TAG}_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find gpurun_out/prof_${TAG}_WRITE_SIZE -name '*counter_collection.csv' | head -1) k_fill_buckets
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
find gpurun_out/prof_${TAG}_trace -name '*kernel_stats.csv' | head -1 | xargs -r head -8
# the default multi-process bench (node-sharded leg = value, replicas beside it): two ranks on this box's one device, the exchange over gloo
KAI_BENCH_BACKEND=gloo KAI_BENCH_ONE_DEVICE=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/${TAG}_multiprocess_rehearsal.json 2> gpurun_out/${TAG}_multiprocess_rehearsal.err; echo "rehearsal rc=$?"
tail -c 1500 gpurun_out/${TAG}_multiprocess_rehearsal.json | cut -c1-700
