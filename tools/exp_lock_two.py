#!/usr/bin/env python3
"""How much of a lock pass is launch latency that another stream could hide: N lock engines (own tables, own streams) replay
the same client trace side by side, 64k batches -- aggregate requests/s against one engine alone.  An upper bound for running
count(k+1) beside resolve(k) inside ONE engine, without building it.
usage: exp_lock_two.py [fasst|2pl] [n_engines] [batches]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dint_amd import wire
from dint_amd.driver import fasst_trace, tpl_trace
from dint_amd.engine import Engine

kind = sys.argv[1] if len(sys.argv) > 1 else "fasst"
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 96
BATCH = 65536
TPL = kind == "2pl"
wl = wire.Workload.TPL if TPL else wire.Workload.FASST
engs = [Engine(wl, n_slots=1 << 20, device=0) for _ in range(NE)]
for e in engs:
    e.snapshot()
stream, recorded, cst = (tpl_trace if TPL else fasst_trace)(engs[0], nb * BATCH, n_workers=4096, key_space=24_000_000, zipf_theta=0.8)
engs[0].sync(); engs[0].restore()
msg = (wire.TPL_MSG if TPL else wire.FASST_MSG).itemsize
d_req = torch.from_numpy(np.frombuffer(stream.tobytes(), np.uint8).copy()).cuda()
d_rep = [torch.empty_like(d_req) for _ in range(NE)]
torch.cuda.synchronize()


def run(n_eng):
    for e in engs[:n_eng]:
        e.restore()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(nb):
        o = b * BATCH * msg
        for k in range(n_eng):
            engs[k].submit_device(d_req.data_ptr() + o, BATCH, d_rep[k].data_ptr() + o, 0)
    for e in engs[:n_eng]:
        e.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / nb * 1e6


out = {"kind": kind, "batches": nb}
for n in range(1, NE + 1):
    run(n)
    us = min(run(n) for _ in range(3))
    out[f"{n}_engines"] = {"us_per_round": round(us, 1), "M_req_s": round(n * BATCH / us, 1)}
    ok = all(d_rep[k].cpu().numpy().tobytes() == recorded.tobytes() for k in range(n))
    out[f"{n}_engines"]["replies_equal_recording"] = ok
print(json.dumps(out))
