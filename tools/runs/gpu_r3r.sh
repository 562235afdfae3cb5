#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; T=${1:-r03r}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fraction or gpu_memory or mig or scan_grid or shared_devices or broad or elastic" > gpurun_out/${T}_pytest.txt 2>&1
tail -3 gpurun_out/${T}_pytest.txt
for w in 1 32; do
  KAI_SCAN_WGS=$w timeout 600 python bench.py --config C3 --fractions 0.3 --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"equal_to_oracle": [a-z]*\|"ops_sha256": "[0-9a-f]*"' | head -3
done
KAI_SCAN_WGS=16 CAMPAIGN_SECONDS=60 timeout 200 python tools/gpu_campaign_mig.py 50000 67000 2>&1 | tail -2
