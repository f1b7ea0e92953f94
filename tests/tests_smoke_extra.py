"""Extra smoke coverage (kept out of __graft_entry__ so it can grow with the engine)."""
import numpy as np

import tracegen
from dint_amd import wire
from dint_amd.engine import Engine
from oracle import oracle as orc


def smoke_extra():
    req = tracegen.tpl_random(2000, seed=6, n_hot=8)
    eng = Engine(wire.Workload.TPL, n_slots=1 << 16, device=0)
    got = eng.submit(req)
    want = orc.TplOracle(1 << 16).replay(req)
    assert got.tobytes() == want.tobytes(), "lock_2pl replies differ from the oracle"

    # tatp: 3 GPU shard servers vs 3 oracle servers under the closed-loop driver, a few epochs
    from dint_amd.driver import Driver
    from dint_amd.replay import ShardGroup

    grp = ShardGroup(wire.Workload.TATP, 5000)
    ora = [orc.TatpOracle(5000) for _ in range(3)]
    d = Driver(wire.Workload.TATP, 2000, 5000, zipf_theta=0.8)
    for _ in range(8):
        req = d.next()
        got = grp.submit(req)
        for s in range(3):
            assert got[s].tobytes() == ora[s].replay(req[s]).tobytes(), "tatp replies differ from the oracle"
        d.consume(got)
