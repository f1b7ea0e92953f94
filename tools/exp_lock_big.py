#!/usr/bin/env python3
"""Phase timeline of the lock tables' big-bin workgroups (DINT_KV_TRACE=1) on the FaSST client trace, one 64k batch at a
time: per batch the number of big bins, and for the slowest workgroup its bin size and the time between stamps
{start, gathered, dominant slot done, sorted, chunks done, end} in microseconds."""
import os
import sys

os.environ["DINT_KV_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dint_amd import wire
from dint_amd.driver import fasst_trace, tpl_trace
from dint_amd.engine import Engine

slots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 12
TPL = os.environ.get("EXP_WL") == "2pl"
eng = Engine(wire.Workload.TPL if TPL else wire.Workload.FASST, n_slots=slots, device=0)
eng.snapshot()
stream, recorded, cst = (tpl_trace if TPL else fasst_trace)(eng, nb * BATCH, n_workers=4096, key_space=24_000_000, zipf_theta=0.8)
eng.sync()
eng.restore()
d_req = torch.from_numpy(np.frombuffer(stream.tobytes(), np.uint8).copy()).cuda()
d_rep = torch.empty_like(d_req)
msg = (wire.TPL_MSG if TPL else wire.FASST_MSG).itemsize
torch.cuda.synchronize()
eng.kv_trace(True)
for b in range(nb):
    o = b * BATCH * msg
    eng.timing_enable(True)
    eng.submit_device(d_req.data_ptr() + o, BATCH, d_rep.data_ptr() + o, 0)
    eng.sync()
    tim = {k: round(v["avg_us"], 1) for k, v in eng.timing_read().items()}
    eng.timing_enable(False)
    _, wg = eng.kv_trace(True)
    act = wg[wg[:, 0] > 0]
    if len(act) == 0:
        print(b, "no big bins")
        continue
    dur = (act[:, 5].astype(np.int64) - act[:, 0].astype(np.int64)) * 0.01
    k = int(np.argmax(dur))
    r = act[k].astype(np.int64)
    ph = [(r[i] - r[0]) * 0.01 if r[i] else float("nan") for i in range(1, 6)]
    t0 = int(act[:, 0].min())
    hp = [(r[i] - r[0]) * 0.01 if r[i] else float("nan") for i in (10, 11, 12)]
    print(f"batch {b}: nbig {int(r[9])}, workgroups {len(act)}; slowest: c {int(r[8])} hot {int(r[13])} total {dur[k]:.1f} us, "
          f"bitmap path: detected {hp[0]:.1f} tables {hp[1]:.1f} answered {hp[2]:.1f}; "
          f"gathered {ph[0]:.1f} hot {ph[1]:.1f} sorted {ph[2]:.1f} chunks {ph[3]:.1f} end {ph[4]:.1f}; "
          f"first start -> last end {(int(act[:, 5].max()) - t0) * 0.01:.1f} us; median wg {np.median(dur):.1f} us; "
          f"sizes top5 {sorted(act[:, 8].tolist())[-5:]}; kernels {tim}")
