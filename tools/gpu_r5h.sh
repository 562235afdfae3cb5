#!/bin/bash
# round 5: k_fill_counts with the one-node fast path of the counting machine — fill tests, C5 / C2 lines with the wavefronts' clocks
TAG=${1:-r05h}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "counts_fill or bucket_fill or full_size_operations or batch_and_sequential" > gpurun_out/${TAG}_pytest_fill.txt 2>&1; echo "pytest fill rc=$?"; tail -2 gpurun_out/${TAG}_pytest_fill.txt
KAI_PROF=1 KAI_BATCH_TRACE=1 KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 900 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench rc=$?"
grep "kai batch (" gpurun_out/${TAG}_bench_c5.err | tail -1 | cut -c1-300
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c5.json')); e=d['config']['engine']; print('C5', round(d['ms_per_step'],2), round(d['value']), e.get('fill_kernel'), 'plan', e.get('plan_ms'), 'fill', e.get('fill_ms'), 'apply', e.get('apply_ms'), d['parity_full']['equal_to_oracle'])"
KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c2.json')); e=d['config']['engine']; print('C2', round(d['ms_per_step'],3), e.get('fill_kernel'), 'fill', e.get('fill_ms'), 'rounds', e.get('rounds'))"
