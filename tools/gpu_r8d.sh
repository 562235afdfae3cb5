#!/bin/bash
# round 6, session 4: the staged job path with the follow step (a run of tasks of one class stays on its node without an index query) — the -m gpu tests of the sequential engine's
# paths, then the shapes it serves: config 3 + 30 % fractions, mixed config 5, config 5 / config 3 with the batch path off (engine_mode 3), each hashed against its pin
TAG=${1:-r08d}; R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q -k "fraction or memory or mig or shared or smoke or scan_grid or engine_mode or synthetic or sequential or full_size" --durations=5 ) > gpurun_out/${TAG}_pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -10 gpurun_out/${TAG}_pytest_subset.txt | cut -c1-200
KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --config C3 --fractions 0.3 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_c3_fractions.json 2> gpurun_out/${TAG}_bench_c3_fractions.err; echo "c3 fractions rc=$?"
KAI_PROF=1 KAI_BENCH_OTHER_SHAPES=0 timeout 600 python bench.py --config C5 --mixed --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_c5_mixed.json 2> gpurun_out/${TAG}_bench_c5_mixed.err; echo "c5 mixed rc=$?"
python - <<PY
import json
for f in ("c3_fractions", "c5_mixed"):
    d = json.loads(open(f"gpurun_out/${TAG}_bench_{f}.json").read().strip().splitlines()[-1])
    print(f, "ms_per_step", round(d["ms_per_step"], 2), "parity", d.get("parity_full", {}).get("equal_to_oracle"), "index queries", d["config"]["engine"].get("index_queries") if "engine" in d.get("config", {}) else None)
PY
grep "kai prof" gpurun_out/${TAG}_bench_c3_fractions.err | tail -1 | cut -c1-300; grep "kai prof" gpurun_out/${TAG}_bench_c5_mixed.err | tail -1 | cut -c1-300
python - <<'PY' 2>&1 | tee gpurun_out/${TAG}_engine_mode3.txt
import sys, time, json, os
sys.path.insert(0, "tests")
import kai_testlib as T
pkg = T.pkg; pins = json.load(open("profiles/full_size_pins.json"))
for name, idx in (("C3", 2), ("C5", 4)):
    snap, cfg, desc = pkg.synth.config(idx, 1.0); cfg.engine_mode = 3
    with pkg.KaiCore(cfg) as core:
        ssn = core.open_session(snap)
        for it in range(2):
            ssn.reset(); t0 = time.perf_counter(); ops = ssn.execute("allocate"); dt = time.perf_counter() - t0; st = ssn.stats()
        got = [(int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) for o in ops]
        print(f"{desc}, batch path off (engine_mode 3): {dt * 1e3:.1f} ms per cycle, {int(st.decisions)} decisions, {int(st.reserved[0])} index queries, operations hash-equal to the oracle's pin: {T.ops_sha256(got) == pins[name]['ops_sha256']}", flush=True)
        ssn.close()
PY
