#!/usr/bin/env python3
"""The stale-bytes report of r04 (NOTEBOOK.md section 1: under rocprofv3 a 2-6 KB range ending on a 4 KB page boundary of one
reply batch held bytes of another batch), taken apart.  Three smallbank engines replay recorded epochs on their streams while
the host thread does what bench.py's recording did with PAGEABLE memory, every result checked against a second copy that
went through page-locked memory only:
  (a) torch: numpy -> .cuda() -> .cpu()                 (the HIP runtime's pageable H2D and D2H, no engine involved)
  (b) dint_submit from / into pageable numpy arrays     (DINT_NO_BOUNCE=1: r04's direct hipMemcpyAsync on the caller's memory;
                                                         default: through the engine's page-locked staging buffers)
usage: [DINT_NO_BOUNCE=1] [rocprofv3 --kernel-trace -d DIR --] stress_pageable.py [seconds] [accounts]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from dint_amd.engine import Engine, Pinned  # noqa: E402
from dint_amd.replay import Replay, ShardGroup  # noqa: E402

T = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
wl = wire.Workload.SMALLBANK
grp = ShardGroup(wl, N)
grp.sync(); grp.snapshot()
drv = Driver(wl, 262144, N, zipf_theta=0.99)
rp, done, host = Replay.recording(drv, grp, 12, keep_host=12)
grp.sync()
msg = grp.msg
# a fourth engine takes the host-path submissions: same population, its own state
eh = Engine(wl, n_rows=N)
eh.populate(N)
eh.sync(); eh.snapshot()
rng = np.random.default_rng(5)
out = {"no_bounce": bool(os.environ.get("DINT_NO_BOUNCE")), "seconds": T, "torch_roundtrips": 0, "torch_bad": [],
       "submit_batches": 0, "submit_bad": []}


def diff_ranges(a, b):
    bad = np.nonzero(a != b)[0]
    if len(bad) == 0:
        return None
    return {"bytes": int(len(bad)), "first": int(bad[0]), "last": int(bad[-1]), "len": int(len(a)),
            "last_mod_4096": int((bad[-1] + 1) % 4096)}


t_end = time.time() + T
e = 0
while time.time() < t_end:
    # keep the three engines busy on their own streams (device-resident replay, asynchronous)
    grp.restore()
    rp.run(grp, 0, len(rp))
    # (a) pageable torch round trips of batch-like sizes while the GPU works
    for _ in range(4):
        n = int(rng.integers(200, 400_000)) * msg
        a = rng.integers(0, 256, n, dtype=np.uint8)
        back = torch.from_numpy(a).cuda().cpu().numpy()
        out["torch_roundtrips"] += 1
        d = diff_ranges(a, back)
        if d and len(out["torch_bad"]) < 8:
            out["torch_bad"].append(d)
    # (b) the recorded host batches of shard server 0 through dint_submit with pageable arrays, against page-locked ones
    req = host[e % len(host)][0][0]
    eh.restore()
    got = eh.submit(req.copy())  # pageable in, pageable out
    eh.restore()
    n = len(req)
    pi, po = Pinned(max(1, n * msg)), Pinned(max(1, n * msg))
    pi.array[:n * msg] = np.frombuffer(req.tobytes(), np.uint8)
    eh.wait(eh.submit_async(pi.ptr, n, po.ptr))
    out["submit_batches"] += 1
    d = diff_ranges(np.frombuffer(got.tobytes(), np.uint8), po.array[:n * msg])
    if d and len(out["submit_bad"]) < 8:
        d["epoch"] = e % len(host)
        out["submit_bad"].append(d)
    pi.close(); po.close()
    e += 1
    grp.sync()
out["n_torch_bad"], out["n_submit_bad"] = len(out["torch_bad"]), len(out["submit_bad"])
print(json.dumps(out))
