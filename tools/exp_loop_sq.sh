#!/bin/bash
# SQ counters (instruction mix, wave cycles, wait cycles) of the closed loop's kernels: tools/exp_loop.py under two
# rocprofv3 --pmc passes (never combined with another trace domain); per kernel the mean per dispatch
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
P2="SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
for k in 1 2; do
  eval "P=\$P$k"
  rm -rf /tmp/sq$k
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d /tmp/sq$k -o sq -- python $R/tools/exp_loop.py > /dev/null 2>&1
  python - /tmp/sq$k <<'PY'
import glob, sqlite3, sys
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    out = {}
    for n, cn, v, k in rows:
        n = n.split("(")[0].replace("void ", "")[:40]
        if "txn" in n or "kv_" in n:
            out.setdefault(n, {})[cn] = (v, k)
    for n, d in sorted(out.items()):
        print(n, {k: round(v[0]) for k, v in sorted(d.items())}, "dispatches", max(v[1] for v in d.values()))
PY
done
