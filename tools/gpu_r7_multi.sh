#!/bin/bash
# rehearsal of bench.py's N > 1 paths on a box with ONE GPU (every rank on device 0, reductions and the sharded group's exchange over gloo: KAI_BENCH_BACKEND=gloo KAI_BENCH_ONE_DEVICE=1;
# RCCL refuses two ranks on one device): the default (replicas, weak scaling) and KAI_BENCH_MULTI=shard (node-sharded group, strong scaling).  Not a scaling measurement: two ranks share one GPU.
TAG=${1:-r07m}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
export KAI_BENCH_BACKEND=gloo KAI_BENCH_ONE_DEVICE=1 KAI_BENCH_OTHER_SHAPES=0
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_2ranks_replicas.json 2> gpurun_out/${TAG}_bench_2ranks_replicas.err; echo "replicas rc=$?"
KAI_BENCH_MULTI=shard timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --config C2 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_2ranks_shard_c2.json 2> gpurun_out/${TAG}_bench_2ranks_shard_c2.err; echo "shard rc=$?"
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_2ranks_replicas.json", "gpurun_out/${TAG}_bench_2ranks_shard_c2.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "n_gpus", d["n_gpus"], "scaling", d["scaling"], "value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 3), "parallelism", d["config"]["parallelism"][:60], "parity", d.get("parity_full", {}).get("equal_to_oracle"), "replicas leg" if "replicas" in d else "")
    except Exception as ex:
        print(f, "unreadable:", ex); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
