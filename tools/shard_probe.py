#!/usr/bin/env python3
"""Node-sharded group with all ranks as threads on ONE GPU (exchange = device-to-device copies): exchange counts and a time bound for DESIGN.md.
usage: shard_probe.py <config C2|C3|C5> <scale> <world> [offers]"""
import ctypes as C, sys, os, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as G
pkg = G._load_pkg()
idx = {"C2": 1, "C3": 2, "C5": 4}[sys.argv[1]]; scale = float(sys.argv[2]); world = int(sys.argv[3]); offers = int(sys.argv[4]) if len(sys.argv) > 4 else 0
import torch
hip = pkg.core._hip_runtime()
snap, cfg, desc = pkg.synth.config(idx, scale)
barrier = threading.Barrier(world); recvs = [None] * world; out = [None] * world
def mk(rank):
    def ag(send, recv, nbytes):
        recvs[rank] = recv; barrier.wait()
        for q in range(world): hip.hipMemcpy(C.c_void_p(recvs[q] + rank * nbytes), C.c_void_p(send), C.c_size_t(nbytes), 3)
        torch.cuda.synchronize(); barrier.wait(); return 0
    return ag
def run(rank):
    core = pkg.KaiCore(cfg, world=world, rank=rank, offers_per_class=offers, allgather=mk(rank))
    ssn = core.open_session(snap)
    for it in range(2):
        ssn.reset(); barrier.wait(); t = time.perf_counter(); ops = ssn.execute("allocate"); dt = time.perf_counter() - t
    st = ssn.stats(); out[rank] = (len(ops), dt, int(st.reserved[4]), int(st.reserved[0]), int(st.decisions), ops)
    ssn.close(); core.destroy()
ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ths]; [t.join() for t in ths]
one = pkg.KaiCore(cfg); s1 = one.open_session(snap); t = time.perf_counter(); ops1 = s1.execute("allocate"); dt1 = time.perf_counter() - t; s1.close(); one.destroy()
same = all(np.array_equal(o[5]["node"], ops1["node"]) and np.array_equal(o[5]["pod"], ops1["pod"]) for o in out)
print(f"{desc}: world {world} offers {offers or 64}: ops {out[0][0]} action {max(o[1] for o in out)*1e3:.1f} ms (all ranks as threads on one GPU) rounds {out[0][2]} exchanges {out[0][3]} decisions {out[0][4]} | one rank {dt1*1e3:.1f} ms | identical to one rank: {same}")
