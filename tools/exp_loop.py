#!/usr/bin/env python3
"""closed loop on the GPU: host issue time against wall time per epoch, one or two client groups
usage: exp_loop.py [clients] [groups]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dint_amd import wire
from dint_amd.driver import GpuDriver
from dint_amd.replay import GpuLoop, ShardGroup
C = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 1_000_000
grp = ShardGroup(wire.Workload.TATP, n, log_entries=8_000_000)
cap = int(0.75 * C / G) + 4096
gds = [GpuDriver(wire.Workload.TATP, C // G, n, cap, first_client=g * (C // G), zipf_theta=0.8) for g in range(G)]
loop = GpuLoop(grp, gds)
loop.epochs(40); loop.sync()
for rep in range(3):
    tx0 = sum(g.stats()["txns"] for g in gds)
    t0 = time.perf_counter()
    loop.epochs(100)
    t1 = time.perf_counter()
    loop.sync()
    t2 = time.perf_counter()
    tx = sum(g.stats()["txns"] for g in gds) - tx0
    print(f"clients {C} groups {G}: host issue {(t1 - t0) * 1e4:.1f} us/round, wall {(t2 - t0) * 1e4:.1f} us/round, {tx / (t2 - t0) / 1e6:.1f} Mtxn/s")
