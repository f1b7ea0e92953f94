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
