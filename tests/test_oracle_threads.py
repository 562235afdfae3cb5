"""The oracle's optional node-scoring threads (the reference scores the nodes of one decision on goroutines, framework/session.go:243-261):
the operations, pod states, node accounting and shares must not depend on the number of workers."""
import numpy as np

import kai_testlib as T


def test_threaded_node_scoring_changes_nothing():
    snap, cfg, _ = T.pkg.synth.config(4, 0.05)  # 3 276 nodes: above the fan-out threshold
    c2 = T.abi.KaiConfig.from_buffer_copy(cfg); c2.reserved[0] = 400
    one = T.Oracle.run(snap, c2, ("allocate",))
    for threads in (2, 5):
        par = T.Oracle.run(snap, c2, ("allocate",), threads=threads)
        assert par.ops == one.ops and par.stmts == one.stmts
        assert np.array_equal(par.pod_status, one.pod_status) and np.array_equal(par.pod_node, one.pod_node)
        for k in one.nodes: assert np.array_equal(par.nodes[k], one.nodes[k]), k
        for k in one.shares_final: assert np.array_equal(par.shares_final[k], one.shares_final[k]), k
    T.Oracle.lib().kai_oracle_set_threads(1)
