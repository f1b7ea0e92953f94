#!/bin/bash
# r06: the closed loop alone, and its kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c3
mkdir -p "$OUT"
cd "$ROOT"
echo "== fused"; timeout 300 python tools/exp_loop.py 2>&1 | tail -3
echo "== nofuse"; DINT_KV_NO_FUSE=1 timeout 300 python tools/exp_loop.py 2>&1 | tail -3
echo "== trace"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o loop -- python "$ROOT/tools/exp_loop.py" > "$OUT/prof.log" 2>&1
cd "$ROOT"
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/prof/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x, grid_y from kernels order by start").fetchall()
rows = rows[-400:-340]
t0 = rows[0][1]; pe = None
for n, s, e, gx, gy in rows:
    nm = n.split('(')[0].replace('void ', '')[:34]
    print(f"{nm:36s} start {(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {((s - pe) / 1e3 if pe else 0):6.1f} grid {gx}x{gy}")
    pe = e
PY
rm -rf "$OUT/prof"
