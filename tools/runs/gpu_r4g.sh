#!/bin/bash
# round 4: the reference's benchmark shapes with the victim actions on 1 / 8 / 32 workgroups (is 32 the right default for small sessions?)
TAG=${1:-r04g}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for g in 1 8 32; do
  echo "== KAI_VICTIM_WGS=$g"
  KAI_VICTIM_WGS=$g timeout 400 python tools/ref_benchmarks.py --max-nodes 200 --iters 2 --out gpurun_out/${TAG}_reference_benchmarks_wgs${g}.json 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"benchmark\"'):
        r = json.loads(l); print(r['benchmark'], r['nodes'], round(r['host_compiled_engine_ms'], 2), round(r.get('mi355x_open_plus_actions_ms', -1), 1))
"
done
