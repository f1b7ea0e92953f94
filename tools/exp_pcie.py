#!/usr/bin/env python3
"""PCIe-inclusive rate of the host path (dint_submit: H2D, one pass, D2H, synchronous) for one tatp shard server."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from dint_amd.replay import ShardGroup  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
grp = ShardGroup(wire.Workload.TATP, 1_000_000)
d = Driver(wire.Workload.TATP, C, 1_000_000, zipf_theta=0.8)
for _ in range(3):
    req = d.next()
    d.consume(grp.submit(req))
req = d.next()
eng = grp.engines[0]
t = time.perf_counter()
for _ in range(5):
    eng.submit(req[0])
dt = (time.perf_counter() - t) / 5
print({"requests": len(req[0]), "ms_per_submit": round(dt * 1e3, 3), "Mreq_s": round(len(req[0]) / dt / 1e6, 1)})
