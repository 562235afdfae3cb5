import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, kai_testlib as T
cfg = T.abi.default_config()
which = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3]
cases = ((4, 0), (0, 10), (1, 1), (3, 200))
for i in which:
    n_nodes, n_pods = cases[i]
    snap = T.pkg.synth.make_snapshot(n_nodes, n_pods, 77, queue_levels=(1, 3), prefill=0.5)
    if snap.n_jobs: snap.arrays["job_queue"][0] = -1
    print("case", i, n_nodes, n_pods, "J", snap.n_jobs, flush=True)
    with T.pkg.KaiCore(cfg) as core:
        ssn = core.open_session(snap); print(" open ok", flush=True)
        ops = ssn.execute("allocate"); print(" exec ok", len(ops), flush=True)
        ssn.close()
print("done")
