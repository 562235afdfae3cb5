#!/bin/bash
# round 5: k_fill_counts publishing per stretch; sweep of the plan's first look-ahead (KAI_BATCH_H0: jobs a leaf offers per round)
TAG=${1:-r05i}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "counts_fill or bucket_fill or full_size_operations" > gpurun_out/${TAG}_pytest_fill.txt 2>&1; echo "pytest fill rc=$?"; tail -1 gpurun_out/${TAG}_pytest_fill.txt
for h in 256 128 64 32; do
  KAI_BATCH_H0=$h KAI_PROF=1 KAI_BATCH_TRACE=1 timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5_h$h.json 2> gpurun_out/${TAG}_bench_c5_h$h.err
  python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c5_h$h.json')); e=d['config']['engine']; print('H0=$h C5', round(d['ms_per_step'],2), 'plan', e.get('plan_ms'), 'fill', e.get('fill_ms'), 'apply', e.get('apply_ms'), 'rounds', e.get('rounds'), d['parity_full']['equal_to_oracle'])"
done
grep "kai batch (" gpurun_out/${TAG}_bench_c5_h256.err | tail -1 | cut -c1-260
