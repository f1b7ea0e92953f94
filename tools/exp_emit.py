#!/usr/bin/env python3
"""time k_txn_emit / k_txn_consume alone (no engines): DINT_TXN_DBG=1 no look-back, 2 no client logic"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dint_amd import wire
from dint_amd.driver import GpuDriver
C = 524288
d = GpuDriver(wire.Workload.TATP, C, 1_000_000, int(0.75 * C), zipf_theta=0.8)
s = torch.cuda.Stream()
for k in range(5):
    d.next(s.cuda_stream); d.consume(s.cuda_stream)
s.synchronize()
t = time.perf_counter()
for k in range(50):
    d.next(s.cuda_stream); d.consume(s.cuda_stream)
s.synchronize()
print(f"dbg={os.environ.get('DINT_TXN_DBG','0')} emit+consume {(time.perf_counter() - t) / 50 * 1e6:.1f} us")
