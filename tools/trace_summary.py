#!/usr/bin/env python3
"""Per-kernel call count / mean / total of a rocprofv3 --kernel-trace rocpd database (and of its memory copies)."""
import glob
import sqlite3
import sys

for db in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    print(db)
    for r in c.execute("select name, count(*), avg(end - start) / 1e3, sum(end - start) / 1e6 from kernels group by name order by 4 desc limit 16"):
        print("  %-60s calls %6d  avg %8.2f us  total %9.2f ms" % (r[0].split("(")[0][:60], r[1], r[2], r[3]))
    try:
        for r in c.execute("select name, count(*), avg(end - start) / 1e3, sum(end - start) / 1e6 from memory_copies group by name order by 4 desc limit 6"):
            print("  copy %-55s calls %6d  avg %8.2f us  total %9.2f ms" % (str(r[0])[:55], r[1], r[2], r[3]))
    except sqlite3.Error as ex:
        print("  (no memory_copies view: %s)" % ex)
