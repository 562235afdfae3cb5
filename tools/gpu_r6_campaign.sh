#!/bin/bash
# round 6: device campaigns of the library with k_fill_levels / k_plan_gather / the new k_plan_scan and k_plan_leaf: random plain-gang clusters through the C ABI against the oracle
# (tools/batch_campaign.py: operations, Statement numbers, states, shares, counters), then the broad campaign (every action, topology, elastic, minruntime, ...)
TAG=${1:-r06x}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "# tools/batch_campaign.py 600000.. gpu"; CAMPAIGN_SECONDS=${BATCH_SECONDS:-300} timeout 700 python tools/batch_campaign.py 600000 700000 gpu 2>&1 | tail -2; } > gpurun_out/${TAG}_batch_campaign_device.txt 2>&1; tail -1 gpurun_out/${TAG}_batch_campaign_device.txt
CAMPAIGN_SECONDS=${BROAD_SECONDS:-150} CAMPAIGN_SECONDS_MIG=30 SEED_BROAD=860000 SEED_MIG=21000 bash tools/gpu_final_campaign.sh ${TAG} 2>&1 | tail -8
