#!/bin/bash
# round 5: control-lane phase clocks (library built with -DKAI_PROF_VICTIM as libkai_core_prof.so) of the ReclaimLargeJobs shapes and of config 4 at 10 %
TAG=${1:-r05b}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/rl_gpu.py <<'PY'
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
python /tmp/rl_gpu.py 200 500 1000 > gpurun_out/${TAG}_rl.txt 2>&1; cat gpurun_out/${TAG}_rl.txt
KAI_CORE_LIB=$R/kai-scheduler_amd/csrc/libkai_core_prof.so KAI_PROF=1 python /tmp/rl_gpu.py 500 > gpurun_out/${TAG}_rl500_prof.txt 2>&1; grep "kai prof\|Reclaim" gpurun_out/${TAG}_rl500_prof.txt | tail -3
KAI_CORE_LIB=$R/kai-scheduler_amd/csrc/libkai_core_prof.so KAI_PROF=1 timeout 600 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_10pct_prof.json 2> gpurun_out/${TAG}_c4_10pct_prof.err; grep "kai prof\|kai victim" gpurun_out/${TAG}_c4_10pct_prof.err | tail -3
