#!/usr/bin/env python3
"""Where a kv pass's chain goes between its kernels: the bench's headline replay under rocprofv3 --kernel-trace, per engine stream
the kernels in order -- durations, and the gaps between the end of one and the start of the next (launch boundaries).
   python tools/pass_gaps.py [tatp|store|smallbank]      -> gpurun_out/dev/pass_gaps_<wl>.txt"""
import glob
import os
import sqlite3
import subprocess
import sys
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "tatp"
    d = "/tmp/prof_gaps"
    subprocess.run(["rm", "-rf", d])
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--legs", "headline", "--steps", "6", "--warmup", "2"] + sys.argv[2:]
    r = subprocess.run(["rocprofv3", "--kernel-trace", "-d", d, "-o", "gaps", "--"] + base, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                       capture_output=True, text=True, timeout=900)
    out = [f"# rc {r.returncode}"]
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        cols = [x[1] for x in c.execute("pragma table_info(kernels)").fetchall()]
        q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
        rows = [(s, e, n.split("(")[0].replace("void ", "").split("<")[0], str(qq)) for s, e, n, qq in c.execute(f"select start, end, name, {q} from kernels")]
        rows.sort()
        byq = defaultdict(list)
        for s, e, n, qq in rows:
            if n.startswith("k_kv_") or n.startswith("k_lock") or n.startswith("k_log"):
                byq[qq].append((s, e, n))
        for qq, ks in sorted(byq.items()):
            if len(ks) < 200:
                continue
            ks = ks[len(ks) // 3:]  # the timed stretch and what follows it
            dur, gap = defaultdict(list), defaultdict(list)
            for a, b in zip(ks, ks[1:]):
                dur[a[2]].append((a[1] - a[0]) / 1e3)
                if b[0] - a[1] < 200e3:  # (not across a host sync)
                    gap[a[2] + " -> " + b[2]].append((b[0] - a[1]) / 1e3)
            out.append(f"queue {qq}: {len(ks)} kernels")
            for n, v in dur.items():
                v = np.array(v)
                out.append(f"   {n:14s} n {len(v):5d}  p10 {np.percentile(v, 10):7.1f} p50 {np.median(v):7.1f} p90 {np.percentile(v, 90):7.1f} mean {v.mean():7.1f} us")
            for n, v in gap.items():
                v = np.array(v)
                out.append(f"   gap {n:30s} n {len(v):5d}  p10 {np.percentile(v, 10):6.1f} p50 {np.median(v):6.1f} p90 {np.percentile(v, 90):6.1f} mean {v.mean():6.1f} us")
    os.makedirs(os.path.join(ROOT, "gpurun_out", "dev"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "dev", f"pass_gaps_{wl}.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
