#!/usr/bin/env python3
"""Experiment: replay the recorded trace with ONE submit per shard server covering all epochs (the C loop enqueues
every pass without returning to Python), one host thread per server -> the GPU-bound rate of the 3 kernel chains."""
import json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire
from dint_amd.driver import Driver
from dint_amd.replay import Replay, ShardGroup, record

n_sub, C, E, W = 1_000_000, 131072, 120, 20
grp = ShardGroup(wire.Workload.TATP, n_sub)
grp.sync(); grp.snapshot()
d = Driver(wire.Workload.TATP, C, n_sub, zipf_theta=0.8)
trace, done = record(d, grp, E)
grp.sync(); grp.restore()
cat_req = [torch.from_numpy(np.frombuffer(b"".join(trace[e][0][s].tobytes() for e in range(E)), np.uint8).copy()).cuda() for s in range(3)]
cat_rep = [torch.empty_like(x) for x in cat_req]
n_all = [sum(len(trace[e][0][s]) for e in range(E)) for s in range(3)]
n_warm = [sum(len(trace[e][0][s]) for e in range(W)) for s in range(3)]
torch.cuda.synchronize()
msg = 55
def sub(s, lo, n):
    grp.engines[s].submit_device(cat_req[s].data_ptr() + lo * msg, n, cat_rep[s].data_ptr() + lo * msg, 0)
for s in range(3): sub(s, 0, n_warm[s])
grp.sync()
for threaded in (False, True):
    grp.restore(); 
    for s in range(3): sub(s, 0, n_warm[s])
    grp.sync()
    t0 = time.perf_counter()
    if threaded:
        th = [threading.Thread(target=sub, args=(s, n_warm[s], n_all[s] - n_warm[s])) for s in range(3)]
        [t.start() for t in th]; [t.join() for t in th]
    else:
        for s in range(3): sub(s, n_warm[s], n_all[s] - n_warm[s])
    t1 = time.perf_counter()
    grp.sync()
    dt = time.perf_counter() - t0
    ops = sum(n_all) - sum(n_warm)
    print(json.dumps({"threaded": threaded, "Mops_s": round(ops / dt / 1e6, 1), "us_per_epoch": round(dt / (E - W) * 1e6, 1), "enqueue_ms": round((t1 - t0) * 1e3, 2), "total_ms": round(dt * 1e3, 2)}))
want = [b"".join(trace[e][1][s].tobytes() for e in range(E)) for s in range(3)]
assert all(cat_rep[s].cpu().numpy().tobytes() == want[s] for s in range(3)), "replies differ"
print("replies identical to the recorded run")
