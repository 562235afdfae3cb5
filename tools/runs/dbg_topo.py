import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, kai_testlib as T
case = T.load_golden("allocate__allocateTopology")["cases"][int(sys.argv[1]) if len(sys.argv) > 1 else 12]
snap, cfg, meta = T.case_to_snapshot(case)
print(meta["name"]); print("pods", snap.pod_names); print("podsets", snap.podset_names, snap.arrays["podset_name_rank"], "pod_podset", snap.arrays["pod_podset"])
ref = T.Oracle.run(snap, cfg)
print("oracle ops", ref.ops)
for mode in (0, 1):
    c = T.abi.KaiConfig.from_buffer_copy(cfg); c.engine_mode = mode
    with T.pkg.KaiCore(c) as core:
        ssn = core.open_session(snap); ops = ssn.execute("allocate"); ssn.close()
    print("gpu mode", mode, [(int(o["kind"]), int(o["pod"]), int(o["node"]), int(o["job"])) for o in ops])
