#!/bin/bash
# round 5: k_fill_counts with the next stretch's parameters prefetched; the counting machine's per-job section clocks (library built with -DKAI_FILL_PROF)
TAG=${1:-r05j}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "counts_fill or bucket_fill or full_size_operations" > gpurun_out/${TAG}_pytest_fill.txt 2>&1; echo "pytest fill rc=$?"; tail -1 gpurun_out/${TAG}_pytest_fill.txt
KAI_PROF=1 KAI_BATCH_TRACE=1 timeout 600 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c5.json')); e=d['config']['engine']; print('C5', round(d['ms_per_step'],2), round(d['value']), 'plan', e.get('plan_ms'), 'fill', e.get('fill_ms'), 'apply', e.get('apply_ms'), 'rounds', e.get('rounds'), d['parity_full']['equal_to_oracle'])"
grep "kai batch (" gpurun_out/${TAG}_bench_c5.err | tail -1 | cut -c1-260
KAI_CORE_LIB=$R/kai-scheduler_amd/csrc/libkai_core_fillprof.so KAI_PROF=1 KAI_BATCH_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5_fillprof.json 2> gpurun_out/${TAG}_bench_c5_fillprof.err
echo "fill-prof build (load = decode, update = decide + emit, rescan = outcome; rescans3 = jobs):"; grep "kai batch (" gpurun_out/${TAG}_bench_c5_fillprof.err | tail -1 | cut -c1-260
