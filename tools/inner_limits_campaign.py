"""Random queue trees whose INNER queues carry GPU limits (synth.make_snapshot(inner_limits_frac=...)): a parent queue turns jobs away that its leaves would take — the gate of an
inner node inside the plan (k_plan_scan pass 1 / k_seg_gate).  Both forms of the plan's scan (KAI_PLAN_SEG_MIN) and both round loops, against the oracle.
usage: inner_limits_campaign.py <seed lo> <seed hi> [gpu]   (default: the kernels on the emulator; CAMPAIGN_SECONDS bounds the run)"""
import sys, os, time; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import kai_testlib as T, numpy as np
GPU = len(sys.argv) > 3 and sys.argv[3] == 'gpu'
if GPU:
    from test_gpu_parity import run_gpu as run
else:
    from test_engine_hostsim import HostSim
    run = HostSim.run
S=T.pkg.synth
lo,hi=int(sys.argv[1]),int(sys.argv[2]); t0=time.time(); bad=tot=0
for seed in range(lo,hi):
    rng=np.random.default_rng(660000+seed)
    snap=S.make_snapshot(int(rng.integers(20,400)), int(rng.integers(200,3000)), 660000+seed, queue_levels=[(2,3),(3,4),(2,2,2),(1,5),(4,)][seed%5], prefill=float(rng.random())*0.6,
                         gpu_mix=[((8,1.0),),((8,.6),(4,.4))][seed%2], zipf=bool(seed%2), limits_frac=0.3 if seed%3==0 else 0.0, inner_limits_frac=(0.5,1.0)[seed%2],
                         gang_sizes=(1,2,4), gang_p=(.6,.3,.1), gpus_per_pod=(1,2) if seed%4 else (1,2,4,8), nonpreempt_frac=0.2*(seed%2), queue_prios=(100,200) if seed%2 else (100,))
    cfg=T.abi.default_config(k_value=(0.0,0.5,1.0)[seed%3], gpu_strategy=T.abi.BINPACK if seed%4 else T.abi.SPREAD)
    os.environ["KAI_PLAN_SEG_MIN"]="1" if seed%3!=2 else "1000000000"
    if seed%4==1: os.environ["KAI_BATCH_HOST_LOOP"]="1"
    else: os.environ.pop("KAI_BATCH_HOST_LOOP",None)
    o=T.Oracle.run(snap,cfg); g=run(snap,cfg); tot+=1
    ok = o.ops==g.ops and (o.pod_status==g.pod_status).all() and (o.pod_node==g.pod_node).all() and all(np.array_equal(o.shares_final[k],g.shares_final[k]) for k in o.shares_final) and (int(o.stats.decisions),int(o.stats.jobs_attempted),int(o.stats.jobs_committed),int(o.stats.rollbacks))==(int(g.stats.decisions),int(g.stats.jobs_attempted),int(g.stats.jobs_committed),int(g.stats.rollbacks)) and int(g.stats.reserved[4])>=1
    if not ok: bad+=1; print("MISMATCH seed",seed, "batch", g.stats.reserved[4], flush=True)
    if time.time()-t0>float(os.environ.get("CAMPAIGN_SECONDS","120")): break
print("inner-limits campaign", "(device)" if GPU else "(host twin)", "runs",tot,"mismatch",bad,f"{time.time()-t0:.0f}s")
