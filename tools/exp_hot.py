import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tracegen
from dint_amd import wire
from dint_amd.engine import Engine
from oracle import oracle as orc
from test_gpu_kv import _hot_tatp
W = wire.Workload
n_sub = 3000
mix = {0: 70, 1: 10, 2: 4, 12: 8, 13: 8}
o = orc.TatpOracle(n_sub, log_entries=400_000)
existing = [o.dump(t)[0] for t in range(5)]
eng = Engine(W.TATP, n_rows=n_sub, log_entries=400_000, flags=int(os.environ.get("FLAGS", "0")))
eng.populate(n_sub)
for k, n in enumerate((6000, 40_000, 150_000)):
    req = _hot_tatp(n, 0.6, mix, seed=10 * k + 3, hot_key=(0, 7), existing=existing, n_noise_sub=n_sub)
    got, want = eng.submit(req), o.replay(req)
    bad = np.nonzero((np.frombuffer(got.tobytes(), "u1").reshape(-1, 55) != np.frombuffer(want.tobytes(), "u1").reshape(-1, 55)).any(axis=1))[0]
    print(k, n, "mismatches", len(bad))
    if len(bad):
        for i in bad[:12]:
            print("  idx", i, "req type", req["type"][i], "table", req["table"][i], "key", hex(int(req["key"][i])), "got", got["type"][i], got["ver"][i], "want", want["type"][i], want["ver"][i])
        hot = (req["table"] == 0) & (req["key"] == 7)
        st = np.nonzero(hot & np.isin(req["type"], [18, 19, 22, 23]))[0]
        print("  structural ops on the hot key at", st[:20], "count", len(st))
        break
