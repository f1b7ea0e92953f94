#!/usr/bin/env python3
"""What bounds an epoch of the replay: the three shard servers' kernel chains (part -> resolve -> big per engine, one stream
each) under variations that need no rebuild.  Prints one JSON object.
  base          the replay as bench.py times it
  skip_big      DINT_EXP_SKIP_BIG=1: the chain without k_kv_big (hot keys unanswered: timing only) = what hiding it could give
  one_engine    one engine alone (no contention)
usage: exp_chain.py [clients] [theta] [tatp|smallbank] [epochs]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from dint_amd.replay import Replay, ShardGroup  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
kind = sys.argv[3] if len(sys.argv) > 3 else "tatp"
E = int(sys.argv[4]) if len(sys.argv) > 4 else 64
wl = {"tatp": wire.Workload.TATP, "smallbank": wire.Workload.SMALLBANK}[kind]
n_rows = 1_000_000 if kind == "tatp" else 10_000_000
grp = ShardGroup(wl, n_rows)
grp.sync(); grp.snapshot()
d = Driver(wl, C, n_rows, zipf_theta=theta if theta > 0 else None)
rp, done, _ = Replay.recording(d, grp, E)
grp.sync(); grp.restore()
torch.cuda.synchronize()
W = E // 4


def timed(engines=(0, 1, 2), reps=3):
    best = None
    for _ in range(reps):
        grp.restore(); grp.sync()
        for e in range(W):
            for s in engines:
                grp.engines[s].submit_device(rp.d_req[e][s], rp.counts[e][s], rp.d_rep[e][s], 0)
        grp.sync()
        t0 = time.perf_counter()
        for e in range(W, E):
            for s in engines:
                grp.engines[s].submit_device(rp.d_req[e][s], rp.counts[e][s], rp.d_rep[e][s], 0)
        grp.sync()
        dt = (time.perf_counter() - t0) / (E - W) * 1e6
        best = dt if best is None else min(best, dt)
    return round(best, 1)


def kernels(engines=(0, 1, 2)):
    grp.restore(); grp.sync()
    for s in engines:
        grp.engines[s].timing_enable(True)
    for e in range(E):
        for s in engines:
            grp.engines[s].submit_device(rp.d_req[e][s], rp.counts[e][s], rp.d_rep[e][s], 0)
    grp.sync()
    t = [grp.engines[s].timing_read() for s in engines]
    for s in engines:
        grp.engines[s].timing_enable(False)
    return {k: round(float(np.mean([x[k]["avg_us"] for x in t])), 1) for k in t[0]}


txn = sum(done[W:E]) / (E - W)
out = {"kind": kind, "clients": C, "theta": theta, "epochs": E, "txn_per_epoch": round(txn),
       "requests_per_epoch": round(rp.ops(W, E) / (E - W))}
out["base"] = {"us_per_epoch": timed(), "kernels_us": kernels()}
out["base"]["Mtxn_s"] = round(txn / out["base"]["us_per_epoch"], 1)
rp.check(0, E)
out["one_engine"] = {"us_per_pass": timed((0,)), "kernels_us": kernels((0,))}
os.environ["DINT_EXP_SKIP_BIG"] = "1"
out["skip_big"] = {"us_per_epoch": timed(), "kernels_us": kernels()}
out["skip_big"]["Mtxn_s_if_hidden"] = round(txn / out["skip_big"]["us_per_epoch"], 1)
out["skip_big_one_engine"] = {"us_per_pass": timed((0,))}
del os.environ["DINT_EXP_SKIP_BIG"]
# same box, same process: r04's hot-key path (kv_big_bin for every big sub, one workgroup per hot key, one kernel)
os.environ["DINT_KV_NO_SPLIT"] = "1"
out["r04_hot_path"] = {"us_per_epoch": timed(), "kernels_us": kernels()}
out["r04_hot_path"]["Mtxn_s"] = round(txn / out["r04_hot_path"]["us_per_epoch"], 1)
out["r04_hot_path_one_engine"] = {"us_per_pass": timed((0,)), "kernels_us": kernels((0,))}
del os.environ["DINT_KV_NO_SPLIT"]
# knobs read at every launch: a sweep costs nothing but the timed replays
if os.environ.get("EXP_SWEEP"):
    sw = {}
    knobs = (("DINT_KV_COARSE_LOAD", ("384", "448", "512", "640", "768")), ("DINT_KV_RPT", ("1", "2")),
             ("DINT_KV_SPLIT_TARGET", ("192", "256", "384", "448")), ("DINT_KV_SPLIT_MIN", ("65", "96", "128", "256")))
    if "=" in os.environ["EXP_SWEEP"]:  # EXP_SWEEP="NAME=v1,v2;NAME2=..." (a value may repeat: the spread of the box)
        knobs = tuple((kv.split("=")[0], tuple(kv.split("=")[1].split(","))) for kv in os.environ["EXP_SWEEP"].split(";"))
    for name, vals in knobs:
        for k, v in enumerate(vals):
            os.environ[name] = v
            try:
                sw[f"{name}={v}" + (f"#{k}" if vals.count(v) > 1 else "")] = timed(reps=2)
            finally:
                del os.environ[name]
    out["sweep_us_per_epoch"] = sw
print(json.dumps(out))
