#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) result directories.

    pmc_summary.py [--last N] <dir> [<dir> ...]

For a --kernel-trace --stats run: the top_kernels view (calls, total/avg duration in us).
For a --pmc run: per kernel and counter, the mean counter value per dispatch.
--last N: also summarise only the last N dispatches of every kernel -- bench.py's event-timed replay is the last thing
that launches kernels, so with N = its launch count these rows are the launches bench.py's `kernels_us` times (the
rows over all calls also contain the population and recording passes).
"""
import glob
import os
import sqlite3
import sys


def short(name: str) -> str:
    return name.split("(")[0].replace("void ", "")[:70]


args = sys.argv[1:]
last = 0
if args and args[0] == "--last":
    last = int(args[1])
    args = args[2:]
for d in args:
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
        if last:
            try:
                names = [r[0] for r in c.execute("select distinct name from kernels")]
                print(f"{'kernel (last %d dispatches)' % last:70s} {'calls':>8s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
                for n in names:
                    du = [r[0] for r in c.execute("select end - start from kernels where name = ? order by start desc limit ?", (n, last))]
                    print(f"{short(n):70s} {len(du):8d} {sum(du) / len(du) / 1e3:10.3f} {min(du) / 1e3:10.3f} {max(du) / 1e3:10.3f}")
            except sqlite3.Error as e:
                print("kernels:", e)
        try:
            rows = c.execute(
                "select kernel_name,counter_name,count(*),avg(value),sum(value),avg(duration) "
                "from counters_collection group by kernel_name,counter_name").fetchall()
            if rows:
                print(f"{'kernel':70s} {'counter':16s} {'dispatches':>10s} {'mean/dispatch':>16s} {'avg_ns':>10s}")
                for n, cn, k, av, sm, du in rows:
                    print(f"{short(n):70s} {cn:16s} {k:10d} {av:16.2f} {du:10.0f}")
            if rows and last:
                print(f"{'kernel (last %d dispatches)' % last:70s} {'counter':16s} {'dispatches':>10s} {'mean/dispatch':>16s} {'avg_ns':>10s}")
                for n, cn in sorted({(r[0], r[1]) for r in rows}):
                    v = c.execute("select value,duration from counters_collection where kernel_name = ? and counter_name = ? "
                                  "order by start desc limit ?", (n, cn, last)).fetchall()
                    print(f"{short(n):70s} {cn:16s} {len(v):10d} {sum(x[0] for x in v) / len(v):16.2f} {sum(x[1] for x in v) / len(v):10.0f}")
        except sqlite3.Error as e:
            print("counters_collection:", e)
