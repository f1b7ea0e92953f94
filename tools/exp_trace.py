#!/usr/bin/env python3
"""Per-wave timeline of k_kv_resolve for one tatp pass (DINT_KV_TRACE=1): where a wave's time goes."""
import os
import sys

os.environ["DINT_KV_TRACE"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from dint_amd.replay import Replay, ShardGroup, record  # noqa: E402

theta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
n_sub, C, E = 1_000_000, 131072, 30
grp = ShardGroup(wire.Workload.TATP, n_sub)
grp.sync(); grp.snapshot()
d = Driver(wire.Workload.TATP, C, n_sub, zipf_theta=theta if theta > 0 else None)
trace, done = record(d, grp, E)
grp.sync(); grp.restore()
rp = Replay(trace, grp.msg)
torch.cuda.synchronize()
eng = grp.engines[0]
import time
per = []
for e in range(E):
    grp.sync(); t0 = time.perf_counter()
    eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], 0)
    grp.sync(); wall = (time.perf_counter() - t0) * 1e6
    tt = eng.kv_trace().astype(np.int64)
    bg = tt[:, 14] > 64
    if bg.any():
        d = (tt[bg, 9] - tt[bg, 8]) / 100
        i = int(np.argmax(d))
        per.append((e, round(wall), int(bg.sum()), int(tt[bg, 14].max()), round(float(d.max()), 1), int(tt[bg][i, 14]), int(tt[bg][i, 13]), int(tt[bg][i, 12])))
print("per pass: (epoch, wall_us, n_big, max_c, slowest_big_us, its_c, its_windows, its_rounds)")
for r in per: print("  ", r)
t = eng.kv_trace().astype(np.int64)
c = t[:, 15]
live = (c > 0) & (t[:, 14] == 0)
t0 = t[live, 0].min()
names = ["entry->c,Skv", "recs", "sort", "msg key+segments", "heads: hdr+locate", "outcomes+replies", "fence+write-back", "rounds+fence", "exit"]
small = live & (c <= 64)
print(f"theta={theta} bins live={live.sum()} small={small.sum()} big={(live & (c > 64)).sum()} max c={c.max()} n={rp.counts[E-1][0]}")
ts = t[small][:, :10]
d_ = np.diff(ts, axis=1)
print("small bins: mean cycles per phase (s_memtime ticks; 100 MHz => x10 ns)")
for k, nm in enumerate(names):
    print(f"  {nm:18s} mean {d_[:, k].mean():9.1f}  p50 {np.median(d_[:, k]):9.1f}  p99 {np.percentile(d_[:, k], 99):9.1f}")
print(f"  wave total        mean {(ts[:, 9] - ts[:, 0]).mean():9.1f}")
rs, re = t[live, 10], t[live, 11]
r0 = rs.min()
print(f"realtime (10 ns ticks): wave start p50 {np.median(rs - r0):.0f} p99 {np.percentile(rs - r0, 99):.0f} max {(rs - r0).max()};  "
      f"wave end p50 {np.median(re - r0):.0f} p90 {np.percentile(re - r0, 90):.0f} p99 {np.percentile(re - r0, 99):.0f} max {(re - r0).max()}")
order = np.argsort(re)[-8:]
print("last waves to finish: (c, start, end)", [(int(c[live][i]), int(rs[i] - r0), int(re[i] - r0)) for i in order])
tl = t[live]
for i in order[-5:]:
    print("   slow wave c=%d phases(cycles):" % tl[i, 15], np.diff(tl[i, :10]).tolist(), "dur_10ns", int(re[i] - rs[i]))
dur = re - rs
sm = tl[:, 15] <= 64
print("small-wave duration (10ns): p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(dur[sm], [50, 90, 99, 100])))
big = t[:, 14] > 64
if big.any():
    tb = t[big]
    o = np.argsort(tb[:, 9] - tb[:, 8])
    print("big bins (c, windows, rounds, dur_us):", [(int(r[14]), int(r[13]), int(r[12]), round((r[9] - r[8]) / 100, 1)) for r in tb[o]])
    # big bins only stamp 0,1 and the per-chunk stamps get overwritten; report count and start
    print("big bins c:", sorted(c[big].tolist())[-10:])
