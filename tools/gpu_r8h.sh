#!/bin/bash
# A/B on one box: the library in the tree against an experimental build (build/libkai_core_exp.so, KAI_CORE_LIB) on the shapes the sequential engine serves
TAG=${1:-r08h}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_OTHER_SHAPES=0 KAI_BENCH_OPEN_LEG=0 KAI_BENCH_NATIVE_FILL=0
for lib in tree exp; do
  if [ $lib = exp ]; then export KAI_CORE_LIB="$R/build/libkai_core_exp.so"; else unset KAI_CORE_LIB; fi
  KAI_PROF=1 timeout 600 python bench.py --config C3 --fractions 0.3 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_${lib}_c3_fractions.json 2> gpurun_out/${TAG}_${lib}_c3_fractions.err
  KAI_PROF=1 timeout 600 python bench.py --config C5 --mixed --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_${lib}_c5_mixed.json 2> gpurun_out/${TAG}_${lib}_c5_mixed.err
  KAI_BENCH_ENGINE_MODE=3 timeout 600 python bench.py --config C3 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_${lib}_c3_mode3.json 2> /dev/null
  KAI_BENCH_ENGINE_MODE=3 timeout 600 python bench.py --config C5 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_${lib}_c5_mode3.json 2> /dev/null
  timeout 300 python bench.py --config C5 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/${TAG}_${lib}_c5.json 2> /dev/null
  timeout 300 python bench.py --config C3 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/${TAG}_${lib}_c3.json 2> /dev/null
  timeout 200 python tools/prof_reclaim.py ReclaimLargeJobs_200 2>/dev/null | tail -1 > gpurun_out/${TAG}_${lib}_reclaim200.txt
  timeout 200 python tools/prof_reclaim.py ConsolidationAction 2>/dev/null | tail -1 > gpurun_out/${TAG}_${lib}_consolidation.txt
done
python - <<PY
import json
for lib in ("tree", "exp"):
    row = [lib]
    for f in ("c3_fractions", "c5_mixed", "c3_mode3", "c5_mode3", "c5", "c3"):
        d = json.loads(open(f"gpurun_out/${TAG}_{lib}_{f}.json").read().strip().splitlines()[-1])
        row.append(f"{f} {d['ms_per_step']:.1f} ms parity {d.get('parity_full', {}).get('equal_to_oracle')}")
    print(" | ".join(row))
    print("   ", open(f"gpurun_out/${TAG}_{lib}_reclaim200.txt").read().strip()[:160]); print("   ", open(f"gpurun_out/${TAG}_{lib}_consolidation.txt").read().strip()[:160])
PY
