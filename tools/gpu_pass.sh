#!/bin/bash
# GPU pass: parity suite + bench lines + kernel trace.  usage: gpu_pass.sh <tag> [full]
TAG=${1:-x}; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_$TAG.log
for spec in "C2 1.0 3" "C3 1.0 2" "C5 0.25 2" "C5 1.0 2"; do set -- $spec
  timeout 400 python bench.py --config $1 --scale $2 --steps $3 --warmup 1 > gpurun_out/bench_${TAG}_$1_$2.log 2>&1; echo "bench $1 x$2 rc=$?"; grep '^{' gpurun_out/bench_${TAG}_$1_$2.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); c=d['config']; print({k:d[k] for k in ('value','ms_per_step')}, {k:c.get(k) for k in ('nodes','pods','decisions_per_step','placements_per_step','session_open_ms','engine')}, d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))"
done
