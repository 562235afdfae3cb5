#!/bin/bash
# round 6, late: long device campaigns of the final library — the batch path's random clusters (a quarter on the host loop, a ninth with short first plans), trees with limited inner queues in
# both forms of the plan's scan, then the broad campaign; everything against the oracle
TAG=${1:-r07c}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "# tools/batch_campaign.py 100000.. gpu"; CAMPAIGN_SECONDS=${BATCH_SECONDS:-400} timeout 900 python tools/batch_campaign.py 100000 200000 gpu 2>&1 | tail -2; } > gpurun_out/${TAG}_batch_campaign_device.txt 2>&1; tail -1 gpurun_out/${TAG}_batch_campaign_device.txt
{ echo "# tools/inner_limits_campaign.py 20000.. gpu"; CAMPAIGN_SECONDS=${INNER_SECONDS:-200} timeout 600 python tools/inner_limits_campaign.py 20000 90000 gpu 2>&1 | tail -2; } > gpurun_out/${TAG}_inner_limits_campaign_device.txt 2>&1; tail -1 gpurun_out/${TAG}_inner_limits_campaign_device.txt
CAMPAIGN_SECONDS=${BROAD_SECONDS:-200} CAMPAIGN_SECONDS_MIG=40 SEED_BROAD=880000 SEED_MIG=23000 bash tools/gpu_final_campaign.sh ${TAG} 2>&1 | tail -8
