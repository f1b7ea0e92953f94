#!/usr/bin/env python3
"""Where the slow smallbank epochs come from: the bench's replay, one epoch at a time, with the engines' timers and counters read
after every epoch; prints the epochs slower than 1.5x the median.  usage: exp_sb_tail.py [epochs] [clients]"""
import os
import sys
import time

os.environ["DINT_KV_TRACE"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from dint_amd.replay import Replay, ShardGroup  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 120
C = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
WL = wire.Workload.SMALLBANK
grp = ShardGroup(WL, 10_000_000, transport="self", n_max=1 << 20)
grp.sync(); grp.snapshot()
drv = Driver(WL, C, 10_000_000, zipf_theta=0.99)
rp, done, _ = Replay.recording(drv, grp, E, inplace=True, ahead=True)
grp.sync(); grp.restore(); rp.reset()
for e in grp.engines:
    e.timing_enable(True)
rows = []
KEYS = ("big_bin_requests", "late_requests", "missing_keys")
for ep in range(E):
    st0 = [e.stats() for e in grp.engines]
    tm0 = [e.timing_read() for e in grp.engines]
    grp.sync(); t0 = time.perf_counter()
    rp.run(grp, ep, ep + 1)
    grp.sync(); wall = (time.perf_counter() - t0) * 1e6
    st1 = [e.stats() for e in grp.engines]
    tm1 = [e.timing_read() for e in grp.engines]
    slow = []
    for e in grp.engines:  # the longest k_kv_big item of the epoch, per engine: {us, records, kind, piece, pieces, records left to kv_big_bin}
        _, wg = e.kv_trace(workgroups=True)
        wg = wg.astype(np.int64)
        if ep == int(os.environ.get("EXP_TIMELINE", "-1")):  # the first item of every k_kv_big workgroup of this pass: in / out since the first one came in
            lv = wg[:, 0] > 0
            t0 = wg[lv, 0].min()
            print("timeline engine %d: %d workgroups, last out %.1f us" % (len(slow), int(lv.sum()), (wg[lv, 1].max() - t0) / 100))
            for r in sorted(wg[lv].tolist(), key=lambda r: r[0])[:: max(1, int(lv.sum()) // 48)]:
                kw = r[30]
                print("   in %6.1f out %6.1f  %-5s %3d/%-3d records %6d" % ((r[0] - t0) / 100, (r[1] - t0) / 100, "SUB PIECE REM SOLO".split()[kw & 3], (kw >> 2) & 255, (kw >> 10) & 255, r[2]))
        w31 = wg[:, 31]
        i = int(np.argmax(w31 >> 40))
        kw = int(w31[i]) & 0x3FFFF
        slow.append({"us": round(int(w31[i] >> 40) / 100, 1), "records": int(w31[i] >> 20) & 0xFFFFF, "kind": "SUB PIECE REM SOLO".split()[kw & 3],
                     "piece": (kw >> 2) & 255, "pieces": (kw >> 10) & 255, "to_big_bin": (int(w31[i]) >> 18) & 1, "workgroups": int((w31 > 0).sum())})
    per = []
    for a, b, ta, tb in zip(st0, st1, tm0, tm1):
        ks = {}
        for k in tb:
            la, lb = ta.get(k, {"launches": 0, "avg_us": 0.0}), tb[k]
            n = lb["launches"] - la["launches"]
            ks[k] = round((lb["avg_us"] * lb["launches"] - la["avg_us"] * la["launches"]) / n, 1) if n else None
        per.append(({k: b[k] - a[k] for k in KEYS}, [b["late_items"][i] - a["late_items"][i] for i in range(3)], ks, slow[len(per)]))
    rows.append((wall, ep, rp.counts[ep], per))
w = np.array([r[0] for r in rows[8:]])
print("epochs %d: wall p50 %.0f p90 %.0f p99 %.0f max %.0f us" % (len(w), np.median(w), np.percentile(w, 90), np.percentile(w, 99), w.max()))
for wall, ep, cnt, per in rows[8:]:
    if wall > 1.5 * np.median(w):
        print("epoch", ep, "wall %.0f" % wall, "batches", list(cnt))
        for k, p in enumerate(per):
            print("   engine", k, p)
