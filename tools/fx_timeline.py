#!/usr/bin/env python3
"""Timeline of one steady-state stretch of `bench.py --force-exchange` under rocprofv3 --kernel-trace (+ memory copies):
per dispatch its start offset, duration and queue -- how the routing kernels, the two exchange copies and the engines'
passes overlap.  Writes gpurun_out/dev/fx_timeline.txt.   python tools/fx_timeline.py [extra bench args]"""
import glob
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d = "/tmp/prof_fx"
    subprocess.run(["rm", "-rf", d])
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-exchange", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-rand64",
            "--no-host-path", "--no-closed-loop", "--no-other-workloads", "--no-shim", "--no-exchange-leg", "--no-as-shipped"] + sys.argv[1:]
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "-d", d, "-o", "fx", "--"] + base, cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
    out = [f"# rc {r.returncode}", "# " + r.stdout.strip().splitlines()[-1][:400] if r.stdout.strip() else "# no stdout"]
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        rows = []
        try:
            cols = [x[1] for x in c.execute("pragma table_info(kernels)").fetchall()]
            out.append("# kernels columns: " + " ".join(cols))
            q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
            rows += [(s, e, n.split("(")[0].replace("void ", "")[:40], str(qq)) for s, e, n, qq in
                     c.execute(f"select start, end, name, {q} from kernels")]
        except sqlite3.Error as ex:
            out.append(f"# kernels: {ex}")
        try:
            cols = [x[1] for x in c.execute("pragma table_info(memory_copies)").fetchall()]
            out.append("# memory_copies columns: " + " ".join(cols))
            rows += [(s, e, "memcpy " + str(n), "copy") for s, e, n in c.execute("select start, end, size from memory_copies")]
        except sqlite3.Error as ex:
            out.append(f"# memory_copies: {ex}")
        rows.sort()
        # the middle of the timed region: 120 dispatches around 60 % of the run
        packs = [k for k, r in enumerate(rows) if "k_route_pack" in r[2]]
        k0 = packs[len(packs) * 2 // 3] if packs else int(len(rows) * 0.6)  # inside the timed region
        t0 = rows[k0][0]
        for s, e, n, qq in rows[k0:k0 + 90]:
            out.append(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  q={qq:8s} {n}")
    os.makedirs(os.path.join(ROOT, "gpurun_out", "dev"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "dev", "fx_timeline.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[:140]))


if __name__ == "__main__":
    main()
