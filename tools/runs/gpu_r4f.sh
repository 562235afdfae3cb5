#!/bin/bash
# round 4: fill v4 + pinned operations buffer: parity subset, bench with step clocks, the reference's own benchmark topologies on the device
TAG=${1:-r04f}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "bucket or batch or synthetic or three_level or full_size_operations" > gpurun_out/${TAG}_pytest_batch.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${TAG}_pytest_batch.txt
KAI_BENCH_TRACE=1 KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --steps 6 --warmup 1 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench rc=$?"; grep -E "bench step" gpurun_out/${TAG}_bench_c5.err | tail -6; grep "kai batch" gpurun_out/${TAG}_bench_c5.err | tail -1
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_bench_c5.json'))
print('C5', d['ms_per_step'], d['value'], d['roofline']['frac'], d['parity_full']['equal_to_oracle'], d.get('cpu_same_algorithm',{}).get('ms_per_step'))
print(json.dumps(d.get('cycle_with_open_ms')), json.dumps(d.get('cycle_pipelined_ms')))
PY
timeout 1200 python tools/ref_benchmarks.py --max-nodes 1000 --no-oracle-above 500 --out gpurun_out/${TAG}_reference_benchmarks.json 2>&1 | cut -c1-330
