#!/usr/bin/env python3
"""Memory-access microbenchmarks of dint_bench_access on the GPU box: the measured denominators of the roofline
fractions bench.py prints (random 64-byte gathers at several table sizes / occupancies, narrow gathers,
read-modify-write, blind scatters, device-scope atomics, streaming).  One JSON object on stdout."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dint_amd.engine import bench_access  # noqa: E402

GB = 1 << 30
N = 1 << 28
out = {"gather64": {}, "other": {}}
for gb in (3, 8, 32):
    for bpc in (4, 8, 16):
        try:
            aps, _ = bench_access(gb * GB, N, "gather", 64, bpc)
            out["gather64"][f"{gb}GB_bpc{bpc}"] = round(aps / 1e9, 3)
        except Exception as ex:  # e.g. out of memory on a shared box
            out["gather64"][f"{gb}GB_bpc{bpc}"] = str(ex)
for name, mode, width, gb in (("gather16_8GB", "gather", 16, 8), ("gather8_8GB", "gather", 8, 8),
                              ("rmw64_8GB", "rmw", 64, 8), ("rmw8_8GB", "rmw", 8, 8),
                              ("scatter1_8GB", "scatter", 1, 8), ("scatter8_8GB", "scatter", 8, 8),
                              ("scatter16_8GB", "scatter", 16, 8), ("scatter64_8GB", "scatter", 64, 8),
                              ("scatter8_4MB", "scatter", 8, 0), ("gather8_16MB", "gather", 8, -1),
                              ("atomic_16MB", "atomic", 8, -1), ("atomic_ret_16MB", "atomic_ret", 8, -1),
                              ("atomic_8GB", "atomic", 8, 8), ("atomic_ret_8GB", "atomic_ret", 8, 8)):
    nbytes = gb * GB if gb > 0 else (4 << 20 if gb == 0 else 16 << 20)
    try:
        aps, _ = bench_access(nbytes, N, mode, width, 8)
        out["other"][name] = round(aps / 1e9, 3)
    except Exception as ex:
        out["other"][name] = str(ex)
for name, mode in (("stream_rd_GBs", "stream_rd"), ("stream_wr_GBs", "stream_wr")):
    aps, _ = bench_access(8 * GB, N, mode, 16, 16)
    out["other"][name] = round(aps * 16 / 1e9, 1)
out["unit"] = "G accesses/s (stream_*: GB/s)"
print(json.dumps(out))
