#!/usr/bin/env python3
"""closed loop on the GPU (GpuDriver + three engines), timed; run under rocprofv3 --kernel-trace --stats for per-kernel times"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dint_amd import wire
from dint_amd.driver import GpuDriver
from dint_amd.replay import GpuLoop, ShardGroup

wl = wire.Workload.TATP if (len(sys.argv) < 2 or sys.argv[1] == "tatp") else wire.Workload.SMALLBANK
C = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
n_rows = 1_000_000 if wl == wire.Workload.TATP else 10_000_000
g = ShardGroup(wl, n_rows)
cap = int(0.75 * C)
d = GpuDriver(wl, C, n_rows, cap, zipf_theta=0.8 if wl == wire.Workload.TATP else 0.99)
loop = GpuLoop(g, d)
loop.epochs(10); loop.sync()
t = time.perf_counter(); tx0 = d.stats()["txns"]
K = 50
t = time.perf_counter()
loop.epochs(K); loop.sync()
dt = time.perf_counter() - t
st = d.stats()
print(f"closed loop {wl.name} clients={C}: {dt / K * 1e3:.4f} ms/epoch  {(st['txns'] - tx0) / dt / 1e6:.1f} Mtxn/s overflow={st['overflow']}")
