#!/usr/bin/env python3
"""One tatp shard server alone on the GPU, no tracing: per-kernel event times and wall time per pass.
usage: exp_pass.py [clients] [theta]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from dint_amd.replay import Replay, ShardGroup, record  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
n_sub, E = 1_000_000, 24
grp = ShardGroup(wire.Workload.TATP, n_sub)
grp.sync(); grp.snapshot()
d = Driver(wire.Workload.TATP, C, n_sub, zipf_theta=theta if theta > 0 else None)
trace, done = record(d, grp, E)
grp.sync(); grp.restore()
rp = Replay(trace, grp.msg)
torch.cuda.synchronize()
eng = grp.engines[0]
for rep in range(2):
    grp.restore(); grp.sync()
    t0 = time.perf_counter()
    for e in range(E):
        eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], 0)
    grp.sync()
    wall = (time.perf_counter() - t0) / E * 1e6
grp.restore()
eng.timing_enable(True)
for e in range(E):
    eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], 0)
grp.sync()
n = float(np.mean([rp.counts[e][0] for e in range(E)]))
print(json.dumps({"clients": C, "theta": theta, "requests_per_pass": round(n), "wall_us_per_pass": round(wall, 1),
                  "Mreq_s_alone": round(n / wall, 1),
                  "kernels_us": {k: round(v["avg_us"], 2) for k, v in eng.timing_read().items()}}))
