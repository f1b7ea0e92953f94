#!/usr/bin/env python3
"""Big-bin statistics of tatp passes (DINT_KV_TRACE=1): per pass, how many bins went to the big-bin workgroups of k_kv_resolve, how many
records they hold, and how long the slowest one took.  usage: exp_big.py [clients] [theta]"""
import os
import sys
import time

os.environ["DINT_KV_TRACE"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from dint_amd.replay import Replay, ShardGroup, record  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
WLN = os.environ.get("EXP_WL", "tatp")  # tatp | smallbank
WL = wire.Workload.SMALLBANK if WLN == "smallbank" else wire.Workload.TATP
n_sub, E = (10_000_000 if WLN == "smallbank" else 1_000_000), int(os.environ.get("EXP_EPOCHS", "16"))
grp = ShardGroup(WL, n_sub)
grp.sync(); grp.snapshot()
d = Driver(WL, C, n_sub, zipf_theta=theta if theta > 0 else None)
trace, done = record(d, grp, E)
grp.sync(); grp.restore()
rp = Replay(trace, grp.msg)
torch.cuda.synchronize()
eng = grp.engines[0]
print("per pass: n, wall_us, n_big, records in big bins, max c, slowest big: us / c / windows / rounds, sum of big us, top-5 c")
for e in range(E):
    grp.sync(); t0 = time.perf_counter()
    eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], 0)
    grp.sync(); wall = (time.perf_counter() - t0) * 1e6
    tt, wgp = eng.kv_trace(workgroups=True)
    tt, wgp = tt.astype(np.int64), wgp.astype(np.int64)
    lv = wgp[:, 0] > 0
    wd = np.where(lv, wgp[:, 1] - wgp[:, 0], 0)
    wi = int(np.argmax(wd))
    print("   slowest workgroup: block %d (%s), %.1f us; kernel span %.1f us" % (
        wi, "big-bin list" if wi < 512 else "bins %d..%d" % ((wi - 512) * 8, (wi - 512) * 8 + 7), wd[wi] / 100,
        (wgp[lv, 1].max() - wgp[lv, 0].min()) / 100), [int(x) for x in tt[(wi - 512) * 8:(wi - 512) * 8 + 8, 15]] if wi >= 512 else "")
    bg = tt[:, 14] > 64
    if bg.any():
        sm = (tt[:, 15] > 0) & (tt[:, 11] > 0)
        t0 = min(tt[sm, 10].min(), tt[bg, 8].min())
        print("     timeline us: small end p50 %.0f max %.0f | big start max %.0f end max %.0f" % (
            np.median(tt[sm, 11] - t0) / 100, (tt[sm, 11] - t0).max() / 100, (tt[bg, 8] - t0).max() / 100, (tt[bg, 9] - t0).max() / 100))
        du = (tt[bg, 9] - tt[bg, 8]) / 100
        i = int(np.argmax(du))
        b = tt[bg]
        print("  ", rp.counts[e][0], round(wall), int(bg.sum()), int(b[:, 14].sum()), int(b[:, 14].max()),
              round(float(du.max()), 1), int(b[i, 14]), int(b[i, 13]), int(b[i, 12]), round(float(du.sum())),
              sorted(b[:, 14].tolist())[-5:], "bins with rounds (c, rounds, us):",
              [(int(r[14]), int(r[12]), round((r[9] - r[8]) / 100, 1)) for r in b if r[12] > 0][:6])
grp.restore()
eng.timing_enable(True)
for e in range(E):
    eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], 0)
grp.sync()
print({k: round(v["avg_us"], 2) for k, v in eng.timing_read().items()})
# timeline of the last traced pass (10 ns ticks of s_memrealtime): big bins and small waves relative to the first start
grp.restore(); grp.sync()
for e in range(E):
    if e == E - 1:
        eng.kv_trace()
    eng.submit_device(rp.d_req[e][0], rp.counts[e][0], rp.d_rep[e][0], 0)
    grp.sync()
tt, wg = eng.kv_trace(workgroups=True)
tt, wg = tt.astype(np.int64), wg.astype(np.int64)
live = wg[:, 0] > 0
w0 = wg[live, 0].min()
dur = (wg[live, 1] - wg[live, 0]) / 100
print("workgroups: %d, start p50 %.1f max %.1f us, end p50 %.1f p99 %.1f max %.1f us" % (
    live.sum(), np.median(wg[live, 0] - w0) / 100, (wg[live, 0] - w0).max() / 100, np.median(wg[live, 1] - w0) / 100,
    np.percentile(wg[live, 1] - w0, 99) / 100, (wg[live, 1] - w0).max() / 100))
idx = np.nonzero(live)[0]
o = np.argsort(wg[live, 1])[-10:]
print("last workgroups out (block, start_us, end_us):", [(int(idx[i]), round((wg[idx[i], 0] - w0) / 100, 1), round((wg[idx[i], 1] - w0) / 100, 1)) for i in o])
small = (tt[:, 15] > 0) & (tt[:, 15] <= 64) & (tt[:, 11] > 0)
s0 = tt[small, 10].min()
bg = (tt[:, 14] > 64) & (tt[:, 8] >= s0 - 100000)
t0 = min(s0, tt[bg, 8].min()) if bg.any() else s0
print("small waves: start p50 %.1f us p99 %.1f | end p50 %.1f p99 %.1f max %.1f" % tuple(
    x / 100 for x in (np.median(tt[small, 10] - t0), np.percentile(tt[small, 10] - t0, 99), np.median(tt[small, 11] - t0),
                      np.percentile(tt[small, 11] - t0, 99), (tt[small, 11] - t0).max())))
rows = sorted(((int(r[14]), round((r[8] - t0) / 100, 1), round((r[9] - t0) / 100, 1), int(r[13]), int(r[12])) for r in tt[bg]), key=lambda r: -r[2])
print("big bins (c, start_us, end_us, stretches, rounds), latest end first:", rows[:12])

names = ["gather", "sort", "A", "B", "C+D", "leaders", "tiles", "wb"]
big_wg = [i for i in range(512) if wg[i, 2] > 0]
for i in sorted(big_wg, key=lambda i: -(wg[i, 1] - wg[i, 0]))[:4] + sorted(big_wg, key=lambda i: wg[i, 1] - wg[i, 0])[:2]:
    if wg[i, 10] > 0:
        hp = [wg[i, 2]] + [wg[i, k] for k in range(10, 16)]
        print("    dominant-key path us (detect+verify, build+sort M, tables, probe, answers, write-back+compact):", [round((hp[k + 1] - hp[k]) / 100, 1) for k in range(6)])
    st = [wg[i, 0]] + [wg[i, k] for k in range(2, 10)]
    print("  workgroup %d, %.1f us in all; first stretch of its bin, phases us:" % (i, (wg[i, 1] - wg[i, 0]) / 100),
          {names[k]: round((st[k + 1] - st[k]) / 100, 1) for k in range(8)})
