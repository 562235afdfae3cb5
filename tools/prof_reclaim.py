"""Phase clocks of the victim search on the reference's benchmark shapes (tools/ref_benchmarks.py): run with KAI_CORE_LIB pointing at a -DKAI_PROF_VICTIM build of the library and
KAI_PROF=1 — the library prints the control lane's cycles per phase (kai_engine_solver.inc KAI_VCLK slots) to stderr.   usage: prof_reclaim.py [name substring]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import kai_testlib as T
import ref_benchmarks as RB
pkg = T.pkg
want = sys.argv[1] if len(sys.argv) > 1 else "ReclaimLargeJobs_200"
for name, build, acts, published in RB.BENCHES:
    if want not in name: continue
    snap, cfg, _ = T.case_to_snapshot(build(), acts)
    with pkg.KaiCore(cfg) as core:
        for it in range(2):
            t0 = time.perf_counter(); ssn = core.open_session(snap); t1 = time.perf_counter()
            n = sum(len(ssn.execute(a)) for a in acts); t2 = time.perf_counter()
            st = ssn.stats(); ssn.close()
            print(f"{name}: nodes {snap.n_nodes} pods {snap.n_pods} open {1e3 * (t1 - t0):.2f} ms actions {1e3 * (t2 - t1):.2f} ms operations {n} scenarios {int(st.reserved[2])} simulations {int(st.reserved[3])}", flush=True)
