import numpy as np, sys
sys.path.insert(0,"tests"); sys.path.insert(0,".")
import tracegen
from dint_amd import wire
from dint_amd.engine import Engine
from oracle import oracle as orc
def run(name, req, flags, n_acct=10000, pop=40):
    eng = Engine(wire.Workload.SMALLBANK, n_rows=n_acct, log_entries=70000, flags=flags); eng.populate(pop)
    o = orc.SmallbankOracle(n_acct, log_entries=70000, populate_n=pop)
    got, want = eng.submit(req), o.replay(req)
    res = [got.tobytes()==want.tobytes()]
    for t in range(2):
        a,b = eng.dump_rows(t), o.dump(t)
        bad = np.nonzero((a[1]!=b[1]) | (a[2]!=b[2]).any(axis=1))[0]
        res.append(len(bad))
        if len(bad) and t == 0:
            i = bad[0]
            k = int(a[0][i])
            sel = (req["key"]==k) & (req["table"]==0)
            print("   key", k, "eng ver", int(a[1][i]), "orc ver", int(b[1][i]), "ops on key:", req["type"][sel].tolist()[:40], "idx", np.nonzero(sel)[0][:12].tolist())
    print(name, "flags", flags, "replies_equal, bad0, bad1 =", res)
req = tracegen.sb_random(5000, seed=5040, n_acct_touch=40)
run("mixed", req, 1); run("mixed", req, 0)
# singles only: every request a distinct account
r2 = tracegen.sb_random(3000, seed=1, n_acct_touch=3000); r2["key"] = np.arange(3000); r2["table"] = 0
run("singles", r2, 0, pop=3000)
# exactly two requests per account, adjacent
r3 = tracegen.sb_random(60, seed=2, n_acct_touch=30); r3["key"] = np.repeat(np.arange(30), 2); r3["table"] = 0
run("pairs", r3, 0, pop=40)
for n in (10, 60, 64, 65, 100, 200):
    r = tracegen.sb_random(n, seed=n, n_acct_touch=1); r["table"] = 0; r["type"] = np.where(r["type"] == 6, 0, r["type"])
    run(f"onekey n={n}", r, 0)
