#!/bin/bash
# round 5: the control lane's larger data out of EngineLocal (k_action's private segment 8 208 -> 4 336 B per lane): the three-rank group that ran out of queue resources, the victim tests, C5 / C4 10 % / ReclaimLargeJobs lines
TAG=${1:-r05n}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "victim_waves_over_the_ranks or node_sharded_group" > gpurun_out/${TAG}_ranks_$i.txt 2>&1; echo "ranks run $i rc=$?"; tail -1 gpurun_out/${TAG}_ranks_$i.txt | cut -c1-120; done
KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c5.json')); e=d['config']['engine']; print('C5', round(d['ms_per_step'],2), round(d['value']), 'plan', e.get('plan_ms'), 'fill', e.get('fill_ms'), d['parity_full']['equal_to_oracle'])"
timeout 600 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_10pct.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_c4_10pct.json')); print('C4 10%', round(d['ms_per_step']/1e3,2), 's')"
timeout 900 python tools/ref_benchmarks.py --max-nodes 1000 --iters 2 --out gpurun_out/${TAG}_reference_benchmarks.json > gpurun_out/${TAG}_reference_benchmarks.log 2>&1; grep -o '"benchmark": "[A-Za-z_0-9]*"\|"mi355x_open_plus_actions_ms": [0-9.]*' gpurun_out/${TAG}_reference_benchmarks.log | paste - - | tail -13 | cut -c14-120
