#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) result directories.

    pmc_summary.py <dir> [<dir> ...]

For a --kernel-trace --stats run: the top_kernels view (calls, total/avg duration in us).
For a --pmc run: per kernel and counter, the mean counter value per dispatch.
"""
import glob
import os
import sqlite3
import sys


def short(name: str) -> str:
    return name.split("(")[0].replace("void ", "")[:70]


for d in sys.argv[1:]:
    for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
        c = sqlite3.connect(db)
        print(f"# {os.path.relpath(db)}")
        try:
            rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
            if rows:
                print(f"{'kernel':70s} {'calls':>8s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
                for n, calls, tot, avg, pct in rows:
                    print(f"{short(n):70s} {calls:8d} {tot:12.2f} {avg:10.3f} {pct:7.2f}")
        except sqlite3.Error as e:
            print("top_kernels:", e)
        try:
            rows = c.execute(
                "select kernel_name,counter_name,count(*),avg(value),sum(value),avg(duration) "
                "from counters_collection group by kernel_name,counter_name").fetchall()
            if rows:
                print(f"{'kernel':70s} {'counter':16s} {'dispatches':>10s} {'mean/dispatch':>16s} {'avg_ns':>10s}")
                for n, cn, k, av, sm, du in rows:
                    print(f"{short(n):70s} {cn:16s} {k:10d} {av:16.2f} {du:10.0f}")
        except sqlite3.Error as e:
            print("counters_collection:", e)
