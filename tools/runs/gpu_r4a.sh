#!/bin/bash
# round 4, first call: BASELINE config 4 at FULL size (10 000 nodes x 110 000 pods, queueDepthPerAction 8 on the victim actions) on the device; its operations must hash to
# profiles/full_size_pins.json C4_100pct_depth8 (oracle end to end, 8 197 s).  One cycle, no warm-up, no CPU legs.
TAG=${1:-r04a}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python bench.py --config C4 --scale 1 --queue-depth 8 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_full.json 2> gpurun_out/${TAG}_c4_full.err; echo "bench rc=$?"
tail -3 gpurun_out/${TAG}_c4_full.err
python - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_c4_full.json'))
print(d['ms_per_step'], d['value'], json.dumps(d['parity_full']), json.dumps(d['config']['engine'])[:600])
PY
