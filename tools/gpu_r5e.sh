#!/bin/bash
# round 5: k_fill_counts after its first tuning (wait / idle clocks of the two wavefronts), whole GPU suite
TAG=${1:-r05e}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
KAI_PROF=1 KAI_BATCH_TRACE=1 KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 900 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench rc=$?"
grep "kai batch (" gpurun_out/${TAG}_bench_c5.err | tail -1 | cut -c1-300
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c5.json')); e=d['config']['engine']; print('C5', round(d['ms_per_step'],2), round(d['value']), e.get('fill_kernel'), 'plan', e.get('plan_ms'), 'fill', e.get('fill_ms'), 'apply', e.get('apply_ms'), d['parity_full']['equal_to_oracle'])"
if [ -f kai-scheduler_amd/csrc/libkai_core_vpop.so ]; then KAI_CORE_LIB=$R/kai-scheduler_amd/csrc/libkai_core_vpop.so KAI_PROF=1 timeout 600 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_10pct_vpop.json 2> gpurun_out/${TAG}_c4_10pct_vpop.err; grep "kai prof\|kai victim" gpurun_out/${TAG}_c4_10pct_vpop.err | tail -2; fi
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$?"; tail -3 gpurun_out/${TAG}_pytest_gpu.txt
