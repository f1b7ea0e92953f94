#!/usr/bin/env python3
"""Experiment: per-kernel time of ONE tatp shard server replaying its recorded batches on one stream
(no overlap with the other two servers), for the Zipf and the reference key distributions."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from dint_amd.replay import Replay, ShardGroup, record  # noqa: E402

n_sub, C, E = 1_000_000, 131072, 60
for theta in (0.8, None, 0.01):
    grp = ShardGroup(wire.Workload.TATP, n_sub)
    grp.sync(); grp.snapshot()
    d = Driver(wire.Workload.TATP, C, n_sub, zipf_theta=theta)
    trace, done = record(d, grp, E)
    grp.sync(); grp.restore()
    rp = Replay(trace, grp.msg)
    torch.cuda.synchronize()
    eng = grp.engines[0]
    st = torch.cuda.current_stream().cuda_stream
    # all three servers must advance (state), but time only server 0 on the torch stream
    eng.timing_enable(True)
    import time
    for e in range(E):  # server 0 alone: the recorded trace makes the three servers independent
        eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], st)
    grp.sync()
    tim = eng.timing_read()
    eng.timing_enable(False)
    eng.restore()
    grp.sync()
    t0 = time.perf_counter()
    for e in range(E):
        eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], st)
    grp.sync()
    wall = (time.perf_counter() - t0) / E * 1e6
    # bin-size distribution of one batch of server 0
    req = trace[E // 2][0][0]
    print(json.dumps({"theta": theta, "n0": rp.counts[E // 2][0], "wall_us_per_pass": round(wall, 1),
                      "kernels_us": {k: round(v["avg_us"], 2) for k, v in tim.items()},
                      "types": {int(t): int((req["type"] == t).sum()) for t in np.unique(req["type"])}}))
    for e in range(E):
        assert rp.d_rep[e][0].cpu().numpy().tobytes() == rp.want[e][0]
    del grp, rp
