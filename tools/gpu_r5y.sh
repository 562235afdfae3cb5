#!/bin/bash
# round 5, last build: device campaigns (batch path incl. k_fill_counts, broad cycles with the victim actions on 32 workgroups, shared GPUs / MIG, config 4 at 2 % and 3 % against the oracle), config 4 at 10 %, the reference's benchmark shapes
TAG=${1:-r05y}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "# tools/batch_campaign.py 70000.. gpu (batch path on the device; even seeds: k_fill_counts where the cluster is plain, odd: one placement per step on k_fill_buckets, every fifth: the general kernel)"; CAMPAIGN_SECONDS=150 timeout 400 python tools/batch_campaign.py 70000 99000 gpu 2>&1 | tail -2; } > gpurun_out/${TAG}_batch_campaign_device.txt 2>&1; cat gpurun_out/${TAG}_batch_campaign_device.txt | tail -2
CAMPAIGN_SECONDS=200 CAMPAIGN_SECONDS_MIG=80 SEED_BROAD=700000 SEED_MIG=9000 bash tools/gpu_final_campaign.sh ${TAG} 2>&1 | tail -12
KAI_PROF=1 timeout 600 python bench.py --config C4 --scale 0.1 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_c4_10pct.json 2> gpurun_out/${TAG}_c4_10pct.err; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_c4_10pct.json')); print('C4 10% (unlimited depth)', round(d['ms_per_step']/1e3,2), 's', d['parity_full'].get('equal_to_oracle'), d['config']['engine'].get('victim_search'))"
timeout 900 python tools/ref_benchmarks.py --max-nodes 1000 --iters 2 --out gpurun_out/${TAG}_reference_benchmarks.json > gpurun_out/${TAG}_reference_benchmarks.log 2>&1; echo "ref benchmarks rc=$?"
grep -o '"benchmark": "[A-Za-z_0-9]*"\|"mi355x_open_plus_actions_ms": [0-9.]*' gpurun_out/${TAG}_reference_benchmarks.log | paste - - | tail -19
