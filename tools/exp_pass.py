#!/usr/bin/env python3
"""One tatp shard server alone on the GPU: per-kernel event times and wall time per pass; with DINT_KV_TRACE=1 also the
phase timeline of the resolve workgroups (10 ns stamps, engine.kv_trace).
usage: [DINT_KV_TRACE=1] exp_pass.py [clients] [theta] [workload]"""
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
wl = {"tatp": wire.Workload.TATP, "smallbank": wire.Workload.SMALLBANK}[sys.argv[3] if len(sys.argv) > 3 else "tatp"]
n_sub, E = (1_000_000 if wl == wire.Workload.TATP else 10_000_000), 24
grp = ShardGroup(wl, n_sub)
grp.sync(); grp.snapshot()
d = Driver(wl, C, n_sub, zipf_theta=theta if theta > 0 else None)
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
if os.environ.get("DINT_KV_TRACE"):
    eng.kv_trace()
eng.timing_enable(True)
for e in range(E):
    eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], 0)
grp.sync()
n = float(np.mean([rp.counts[e][0] for e in range(E)]))
out = {"clients": C, "theta": theta, "requests_per_pass": round(n), "wall_us_per_pass": round(wall, 1),
       "Mreq_s_alone": round(n / wall, 1), "kernels_us": {k: round(v["avg_us"], 2) for k, v in eng.timing_read().items()}}
if os.environ.get("DINT_KV_TRACE"):
    tr, big = eng.kv_trace(workgroups=True)
    tr, big = tr.astype(np.int64), big.astype(np.int64)
    big = big[big[:, 0] > 0]
    if len(big):
        du = (big[:, 1] - big[:, 0]) / 100.0
        top = np.argsort(-du)[:8]
        out["big_subs"] = {"n": int(len(big)), "us_mean": round(float(du.mean()), 1), "us_max": round(float(du.max()), 1),
                           "longest": [[int(big[i, 2]), round(float(du[i]), 1), int(big[i, 30]) & 3] for i in top]}
        out["big_subs"]["slowest_items"] = [{"sub": int(big[i, 2]), "us": round(float(du[i]), 1), "kind": int(big[i, 30]) & 3, "closed_form": int(big[i, 30] >> 16) & 1, "chunks_only": int(big[i, 30] >> 17) & 1, "structural": int(big[i, 30] >> 18) & 1, "phased": int(big[i, 30] >> 19) & 1,
                                             "hot": int(big[i, 30] >> 20) & 0xFFF, "rem": int(big[i, 30] >> 32) & 0xFFFFFF, "why_not": int(big[i, 30] >> 56) & 31} for i in np.argsort(-du)[:16]]
        # the first work item of every workgroup by kind (0 whole sub, 1 piece of a hot key, 2 remainder, 3 solo) and sub size
        kind = big[:, 30] & 3
        out["big_subs"]["by_kind_n_mean_max_us"] = {int(k): [int((kind == k).sum()), round(float(du[kind == k].mean()), 1), round(float(du[kind == k].max()), 1)]
                                                    for k in sorted(set(kind.tolist()))}
        sz = big[:, 2]
        out["big_subs"]["by_size_n_mean_max_us"] = {f"{lo}-{hi}": [int(m.sum()), round(float(du[m].mean()), 1), round(float(du[m].max()), 1)]
                                                    for lo, hi in ((65, 128), (128, 256), (256, 512), (512, 1024), (1024, 2048), (2048, 8192))
                                                    for m in [(sz >= lo) & (sz < hi)] if m.any()}
        # the stamped stretch of the longest sub (kv_big_bin: [4] in .. [13] out)
        bn = ["gathered", "ordered", "heads", "keys", "masks", "located_granted", "tiles", "written", "out"]
        i = int(top[0])
        out["big_subs"]["stretch_us"] = {bn[k - 5]: round(float(big[i, k] - big[i, k - 1]) / 100.0, 2) for k in range(5, 14) if big[i, k] > 0 and big[i, k - 1] > 0}
        out["big_subs"]["stretches_rounds"] = [int(big[i, 15]), int(big[i, 14])]
        hn = ["sampled", "checked", "ops_sorted", "row_located", "answered", "compacted"]
        out["big_subs"]["pf_regrouped_nwin_t0gather_us"] = [int(big[i, 27]), int(big[i, 28]), int(big[i, 29]), round(float(big[i, 31] - big[i, 4]) / 100.0, 2)]
        out["big_subs"]["dominant_key_bad_ops_hot_m"] = [int(big[i, k]) for k in range(23, 27)]
        out["big_subs"]["dominant_key_us"] = {hn[k - 17]: round(float(big[i, k] - big[i, k - 1]) / 100.0, 2) for k in range(17, 23) if big[i, k] > 0 and big[i, k - 1] > 0}
    tr = tr[tr[:, 0] > 0]
    t0 = tr[:, 0].min()
    names = ["in", "loaded", "counted", "laid_out", "placed", "sorted", "masks", "located", "replied", "written", "rounds", "chunks_done", "bigs_done"]
    ph = {}
    for k in range(1, 13):
        ok = (tr[:, k] > 0) & (tr[:, k - 1] > 0)
        dd = (tr[ok, k] - tr[ok, k - 1]) / 100.0
        if len(dd):
            ph[names[k]] = [round(float(dd.mean()), 2), round(float(np.percentile(dd, 95)), 2), round(float(dd.max()), 2)]
    out["phases_us_mean_p95_max"] = ph
    start = (tr[:, 0] - t0) / 100.0
    end = (tr[:, 12] - t0) / 100.0
    out["wg_start_us_max"] = round(float(start.max()), 2)
    out["wg_end_us_mean_max"] = [round(float(end.mean()), 2), round(float(end.max()), 2)]
    order = np.argsort(-(tr[:, 12] - tr[:, 0]))[:6]
    out["longest_wgs"] = [{"bin": int(tr[i, 13]), "records": int(tr[i, 14]), "in_big_subs": int(tr[i, 15]), "chunks": int(tr[i, 16]),
                           "start": round(float(start[i]), 1), "chunks_us": round(float(tr[i, 11] - tr[i, 0]) / 100.0, 1),
                           "bigs_us": round(float(tr[i, 12] - tr[i, 11]) / 100.0, 1)} for i in order]
    out["records_per_wg_mean_max"] = [round(float(tr[:, 14].mean()), 1), int(tr[:, 14].max())]
    out["wgs_with_big_subs"] = int((tr[:, 15] > 0).sum())
print(json.dumps(out))
