#!/bin/bash
# final-build campaign on one MI355X: broad randomized cycles, shared-GPU / GPU-memory / MIG cycles, and BASELINE config 4 at 2 % and 3 % against the oracle
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; OUT=gpurun_out/${1:-r02k}_gpu_campaign.txt
{
echo "# tools/gpu_final_campaign.sh on one MI355X, library of the last commit: operations, pod states, node accounting, shares against the oracle"
echo "## tools/gpu_campaign.py ${SEED_BROAD:-300000}.. (broad cases, every action, engine modes)"
CAMPAIGN_SECONDS=${CAMPAIGN_SECONDS:-270} timeout 400 python tools/gpu_campaign.py ${SEED_BROAD:-300000} $(( ${SEED_BROAD:-300000} + 100000 )) 2>&1 | tail -3
echo "## tools/gpu_campaign_mig.py ${SEED_MIG:-3000}.. (fractions, gpu-memory requests, MIG)"
CAMPAIGN_SECONDS=${CAMPAIGN_SECONDS_MIG:-120} timeout 300 python tools/gpu_campaign_mig.py ${SEED_MIG:-3000} $(( ${SEED_MIG:-3000} + 17000 )) 2>&1 | tail -3
echo "## BASELINE config 4 (allocate, consolidation, reclaim in one session) at 2 % and 3 %"
timeout 300 python - <<'PY'
import sys, os, time
sys.path.insert(0, "tests")
import numpy as np
import kai_testlib as T
from test_gpu_parity import run_gpu
for scale in (0.02, 0.03):
    snap, cfg, desc = T.pkg.synth.config(3, scale)
    acts = ("allocate", "consolidation", "reclaim")
    t = time.time(); o = T.Oracle.run(snap, cfg, acts); to = time.time() - t
    t = time.time(); g = run_gpu(snap, cfg, acts); tg = time.time() - t
    ok = o.ops == g.ops and (o.pod_status == g.pod_status).all() and (o.pod_node == g.pod_node).all() and all(np.array_equal(o.nodes[k], g.nodes[k]) for k in o.nodes) \
        and all(np.array_equal(o.shares_final[k], g.shares_final[k]) for k in o.shares_final)
    print(f"{desc}: {len(o.ops)} operations ({sum(1 for x in o.ops if x[0] == 2)} evictions), identical to the oracle: {ok}; oracle {to:.1f} s, MI355X {tg:.1f} s incl. session open")
PY
} > $OUT 2>&1
cat $OUT
