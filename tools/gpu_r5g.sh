#!/bin/bash
# round 5: k_fill_counts with two set workers; the reference's default cycle on config 5's shape at 0.5 %; config 4 at full size (one cycle through bench.py); ReclaimLargeJobs shapes
TAG=${1:-r05g}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "counts_fill or bucket_fill or full_size_operations or batch_and_sequential or default_cycle" > gpurun_out/${TAG}_pytest_fill.txt 2>&1; echo "pytest fill rc=$?"; tail -2 gpurun_out/${TAG}_pytest_fill.txt
KAI_PROF=1 KAI_BATCH_TRACE=1 KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 900 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err; echo "bench rc=$?"
grep "kai batch (" gpurun_out/${TAG}_bench_c5.err | tail -1 | cut -c1-300
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c5.json')); e=d['config']['engine']; print('C5', round(d['ms_per_step'],2), round(d['value']), e.get('fill_kernel'), 'plan', e.get('plan_ms'), 'fill', e.get('fill_ms'), 'apply', e.get('apply_ms'), d['parity_full']['equal_to_oracle'])"
KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0 timeout 300 python bench.py --config C2 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/${TAG}_bench_c2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c2.json')); e=d['config']['engine']; print('C2', round(d['ms_per_step'],3), e.get('fill_kernel'), 'fill', e.get('fill_ms'), 'rounds', e.get('rounds'))"
KAI_PROF=1 timeout 600 python bench.py --config C5 --scale 0.005 --actions allocate,consolidation,reclaim,preempt --queue-depth 8 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c5_default_cycle_0.5pct.json 2> gpurun_out/${TAG}_c5_default_cycle_0.5pct.err; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_c5_default_cycle_0.5pct.json')); print('C5 x 0.005 default cycle', round(d['ms_per_step'],1), 'ms', d['parity_full'])"
python /tmp/rl_gpu.py 200 500 1000 > gpurun_out/${TAG}_rl.txt 2>&1 || { cat > /tmp/rl_gpu.py <<'PY'
import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import kai_testlib as T
import ref_benchmarks as RB
for n in [int(x) for x in sys.argv[1:]]:
    snap, cfg, _ = T.case_to_snapshot(RB.reclaim_large(n), ("reclaim",))
    with T.pkg.KaiCore(cfg) as core:
        for it in range(2):
            t0 = time.perf_counter(); ssn = core.open_session(snap); ops = ssn.execute("reclaim"); dt = time.perf_counter() - t0; ssn.close()
            print(f"ReclaimLargeJobs {n} nodes: {dt * 1e3:.1f} ms, {len(ops)} operations", flush=True)
PY
python /tmp/rl_gpu.py 200 500 1000 > gpurun_out/${TAG}_rl.txt 2>&1; }; cat gpurun_out/${TAG}_rl.txt
KAI_PROF=1 timeout 900 python bench.py --config C4 --scale 1.0 --queue-depth 8 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_full_depth8.json 2> gpurun_out/${TAG}_c4_full_depth8.err; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_c4_full_depth8.json')); print('C4 full depth 8', round(d['ms_per_step']/1e3,1), 's', d['parity_full']['equal_to_oracle'], d['config']['engine'].get('victim_search'))"
