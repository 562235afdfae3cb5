#!/bin/bash
# round 5, LAST build: device campaigns once more (the victim path changed after r05y: victims queue filled best first, index loops from 48 entries on the scan lanes, EngineBig)
TAG=${1:-r05x}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "# tools/batch_campaign.py 100000.. gpu"; CAMPAIGN_SECONDS=60 timeout 300 python tools/batch_campaign.py 100000 130000 gpu 2>&1 | tail -2; } > gpurun_out/${TAG}_batch_campaign_device.txt 2>&1; tail -1 gpurun_out/${TAG}_batch_campaign_device.txt
CAMPAIGN_SECONDS=210 CAMPAIGN_SECONDS_MIG=60 SEED_BROAD=800000 SEED_MIG=19000 bash tools/gpu_final_campaign.sh ${TAG} 2>&1 | tail -10
