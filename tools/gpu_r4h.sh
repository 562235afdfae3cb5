#!/bin/bash
# round 4: (1) the victim actions' waves over the ranks of a group, handles on threads of one process; (2) capacity prediction + H0 256 of the batch path: parity subset,
# bench C5 / C2 / C3 with round traces; (3) the same group as two PROCESSES on this one device over gloo (bench --config C4 --gpus 2), against the one-process run
TAG=${1:-r04h}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -q -x -k "victim_waves_over_the_ranks" > gpurun_out/${TAG}_pytest_victim_group.txt 2>&1; echo "pytest victim group rc=$?"; tail -3 gpurun_out/${TAG}_pytest_victim_group.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bucket or batch or synthetic or three_level or full_size_operations or node_sharded" > gpurun_out/${TAG}_pytest_batch.txt 2>&1; echo "pytest batch rc=$?"; tail -2 gpurun_out/${TAG}_pytest_batch.txt
KAI_BATCH_TRACE=1 KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --steps 6 --warmup 1 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench C5 rc=$?"
grep "kai batch round" gpurun_out/${TAG}_bench_c5.err | tail -40 > gpurun_out/${TAG}_c5_plan_rounds.txt; grep "kai batch" gpurun_out/${TAG}_bench_c5.err | grep -v round | tail -1
for cfg in C2 C3; do
  KAI_BATCH_TRACE=1 timeout 300 python bench.py --config $cfg --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/${TAG}_bench_${cfg}.json 2> gpurun_out/${TAG}_bench_${cfg}.err; echo "bench $cfg rc=$?"
  KAI_BATCH_NO_CAPACITY=1 KAI_BATCH_H0=16 KAI_BENCH_OPEN_LEG=0 timeout 300 python bench.py --config $cfg --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/${TAG}_bench_${cfg}_before.json 2>/dev/null
done
python - <<PY
import json
for n in ("c5", "C2", "C2_before", "C3", "C3_before"):
    try:
        d = json.load(open('gpurun_out/${TAG}_bench_%s.json' % n))
        print(n, round(d['ms_per_step'], 3), round(d['value']), d.get('parity_full', {}).get('equal_to_oracle'), (d.get('cpu_same_algorithm') or {}).get('ms_per_step'), d['config']['engine'].get('rounds'))
    except Exception as e: print(n, 'failed', e)
PY
# the group as two processes on this one device (gloo; every rank on device 0): BASELINE config 4 at 2 %, victim waves over the ranks; beside it the one-process run
KAI_PROF=1 timeout 200 python bench.py --config C4 --scale 0.02 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_one_process.json 2> gpurun_out/${TAG}_c4_one_process.err; echo "C4 one process rc=$?"; grep "kai victim" gpurun_out/${TAG}_c4_one_process.err | tail -2
KAI_PROF=1 KAI_BENCH_BACKEND=gloo KAI_BENCH_ONE_DEVICE=1 KAI_BENCH_REPLICAS_LEG=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --config C4 --scale 0.02 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_two_processes.json 2> gpurun_out/${TAG}_c4_two_processes.err; echo "C4 two processes rc=$?"; grep "kai victim" gpurun_out/${TAG}_c4_two_processes.err | tail -4
python - <<PY
import json
for n in ("one_process", "two_processes"):
    try:
        d = json.loads(open('gpurun_out/${TAG}_c4_%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['ms_per_step'], 1), d['config']['placements_per_step'], json.dumps(d['config']['engine'].get('victim_search')))
    except Exception as e: print(n, 'failed', e)
PY
