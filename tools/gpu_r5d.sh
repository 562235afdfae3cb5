#!/bin/bash
# round 5: the fill as two wavefronts on the MI355X (tests, the default bench line), the victim search after the round's changes, the bucket fill natively on the box's host core
TAG=${1:-r05d}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "counts_fill or bucket_fill or full_size_operations or batch_and_sequential" > gpurun_out/${TAG}_pytest_fill.txt 2>&1; echo "pytest fill rc=$?"; tail -3 gpurun_out/${TAG}_pytest_fill.txt
KAI_PROF=1 KAI_BATCH_TRACE=1 KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_c5.json'))
e=d['config']['engine']
print('C5', round(d['ms_per_step'],2), round(d['value']), e.get('fill_kernel'), 'plan', e.get('plan_ms'), 'fill', e.get('fill_ms'), 'apply', e.get('apply_ms'), 'rounds', e.get('rounds'), d['parity_full']['equal_to_oracle'])
print(json.dumps(d.get('cpu_same_algorithm'))[:900])
print(json.dumps(d['roofline'].get('own_roofline')))
PY
KAI_FILL_ONE_WAVE=1 KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 600 python bench.py --steps 6 --warmup 2 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5_one_wave.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c5_one_wave.json')); e=d['config']['engine']; print('C5 one-wave kernel', round(d['ms_per_step'],2), e.get('fill_kernel'), 'fill', e.get('fill_ms'))"
KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c2.json')); e=d['config']['engine']; print('C2', round(d['ms_per_step'],3), e.get('fill_kernel'), 'fill', e.get('fill_ms'), 'rounds', e.get('rounds'))"
bash tools/gpu_r5b.sh ${TAG}
timeout 900 python tools/native_fill_timing.py --config C5 --scale 1.0 --out gpurun_out/${TAG}_native_fill_c5.json | cut -c1-600
