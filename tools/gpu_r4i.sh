#!/bin/bash
# round 4: k_fill_buckets places a one-class gang in steps of whole nodes — parity (bucket / batch tests, the full-size hash), bench C5 batched vs one placement per step, C2
TAG=${1:-r04i}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bucket or batch or synthetic or three_level or full_size_operations" > gpurun_out/${TAG}_pytest_batch.txt 2>&1; echo "pytest batch rc=$?"; tail -2 gpurun_out/${TAG}_pytest_batch.txt
KAI_BATCH_TRACE=1 KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --steps 6 --warmup 1 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench C5 rc=$?"
grep "kai batch round" gpurun_out/${TAG}_bench_c5.err | tail -12 > gpurun_out/${TAG}_c5_plan_rounds.txt; grep "kai batch" gpurun_out/${TAG}_bench_c5.err | grep -v round | tail -1
KAI_FILL_UNBATCHED=1 KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 timeout 600 python bench.py --steps 6 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5_unbatched.json 2>/dev/null; echo "bench C5 unbatched rc=$?"
timeout 300 python bench.py --config C2 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/${TAG}_bench_C2.json 2>/dev/null; echo "bench C2 rc=$?"
python - <<PY
import json
for n in ("c5", "c5_unbatched", "C2"):
    try:
        d = json.load(open('gpurun_out/${TAG}_bench_%s.json' % n))
        print(n, round(d['ms_per_step'], 3), round(d['value']), d.get('parity_full', {}).get('equal_to_oracle'), (d.get('cpu_same_algorithm') or {}).get('ms_per_step'), d['config']['engine'].get('rounds'), d['roofline'].get('avg_launch_ms'), d['config']['engine'].get('fill_ms'))
    except Exception as e: print(n, 'failed', e)
PY
