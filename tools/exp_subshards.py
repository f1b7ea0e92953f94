#!/usr/bin/env python3
"""Experiment: S hash sub-shards per logical shard server on one GPU (3*S engines, 3*S streams), clients scaled
with S so every engine still sees ~61k-request passes.  Measures whole-GPU request rate of the recorded replay."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire
from dint_amd.driver import Driver
from dint_amd.engine import Engine

S = int(sys.argv[1]); n_sub = 1_000_000; C = 131072 * S; E = 40; W = 10
msg = 55
engs = [[Engine(wire.Workload.TATP, n_rows=n_sub, shard_index=j, shard_count=S) for j in range(S)] for _ in range(3)]
for row in engs:
    for e in row: e.populate(n_sub)
for row in engs:
    for e in row: e.sync(); e.snapshot()
d = Driver(wire.Workload.TATP, C, n_sub, zipf_theta=0.8)
trace = []
for ep in range(E):
    req = d.next(); rep = []
    parts = []
    for s in range(3):
        r = req[s]; n = len(r)
        dr = torch.from_numpy(np.frombuffer(r.tobytes(), np.uint8).copy()).cuda()
        home = torch.empty(n, dtype=torch.uint8, device="cuda")
        engs[s][0].home_shard(dr, n, home); torch.cuda.synchronize()
        h = home.cpu().numpy()
        out = r.copy(); sub = []
        for j in range(S):
            sel = np.nonzero(h == j)[0]
            got = engs[s][j].submit(r[sel]) if len(sel) else r[sel]
            out[sel] = got; sub.append(r[sel])
        rep.append(out); parts.append(sub)
    d.consume(rep); trace.append(parts)
for row in engs:
    for e in row: e.sync(); e.restore()
dev = [[[torch.from_numpy(np.frombuffer(trace[ep][s][j].tobytes(), np.uint8).copy()).cuda() for j in range(S)] for s in range(3)] for ep in range(E)]
out = [[[torch.empty_like(dev[ep][s][j]) for j in range(S)] for s in range(3)] for ep in range(E)]
torch.cuda.synchronize()
def run(lo, hi):
    for ep in range(lo, hi):
        for s in range(3):
            for j in range(S):
                n = len(trace[ep][s][j])
                if n: engs[s][j].submit_device(dev[ep][s][j], n, out[ep][s][j], 0)
run(0, W)
for row in engs:
    for e in row: e.sync()
t0 = time.perf_counter(); run(W, E)
for row in engs:
    for e in row: e.sync()
dt = time.perf_counter() - t0
ops = sum(len(trace[ep][s][j]) for ep in range(W, E) for s in range(3) for j in range(S))
print(json.dumps({"subshards": S, "engines": 3 * S, "clients": C, "Mops_s": round(ops / dt / 1e6, 1), "us_per_epoch": round(dt / (E - W) * 1e6, 1), "reqs_per_engine_pass": ops // ((E - W) * 3 * S)}))
