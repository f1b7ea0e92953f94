"""GPU parity under realistic traffic: the closed-loop TATP / SmallBank drivers against three GPU shard
servers (dint_amd.replay.ShardGroup) in lock step with three CPU oracle servers -- every reply of every
epoch must be identical -- and the record / replay machinery bench.py times."""
import numpy as np
import pytest

from dint_amd import wire
from dint_amd.driver import Driver
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
W = wire.Workload


def _lockstep(workload, n_rows, clients, epochs, zipf, mk_oracle):
    from dint_amd.replay import ShardGroup

    grp = ShardGroup(workload, n_rows, log_entries=200_000)
    ora = [mk_oracle() for _ in range(3)]
    d = Driver(workload, clients, n_rows, zipf_theta=zipf)
    for e in range(epochs):
        req = d.next()
        got = grp.submit(req)
        want = [ora[s].replay(req[s]) for s in range(3)]
        for s in range(3):
            assert got[s].tobytes() == want[s].tobytes(), (e, s)
        d.consume(got)
    return grp, ora, d


@pytest.mark.parametrize("n_sub,clients,zipf", [(20_000, 6000, None), (20_000, 6000, 0.8), (300, 5000, None)])
def test_tatp_closed_loop_gpu_equals_oracle(n_sub, clients, zipf):
    grp, ora, d = _lockstep(W.TATP, n_sub, clients, 50, zipf, lambda: orc.TatpOracle(n_sub, log_entries=200_000))
    for s in range(3):
        assert ora[s].errors == 0
        st = grp.engines[s].stats()
        assert st["bad_requests"] == 0 and st["missing_keys"] == 0 and st["pool_exhausted"] == 0
        for t in range(5):
            a, b = grp.engines[s].dump_rows(t), ora[s].dump(t)
            assert all((x == y).all() for x, y in zip(a, b)), (s, t)
            lk, _ = grp.engines[s].read_locks(t)
            assert (lk == ora[s].locks(t)).all()
        ring, tail = grp.engines[s].read_log(200_000)
        assert tail == ora[s].tail and (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64) == ora[s].ring).all()
    assert d.stats()["committed"] > 0


@pytest.mark.parametrize("n_acct,clients,zipf", [(600_000, 4000, None), (50_000, 4000, 0.99 - 1e-9), (2000, 3000, None)])
def test_smallbank_closed_loop_gpu_equals_oracle(n_acct, clients, zipf):
    grp, ora, d = _lockstep(W.SMALLBANK, n_acct, clients, 50, zipf,
                            lambda: orc.SmallbankOracle(n_acct, log_entries=200_000))
    for s in range(3):
        assert ora[s].errors == 0
        for t in range(2):
            ex, sh = grp.engines[s].read_locks(t)
            assert (ex == ora[s].num_ex(t)).all() and (sh == ora[s].num_sh(t)).all()
            a, b = grp.engines[s].dump_rows(t), ora[s].dump(t)
            assert all((x == y).all() for x, y in zip(a, b)), (s, t)


def test_record_then_replay_from_hbm_is_bit_identical():
    import torch

    from dint_amd.replay import Replay, ShardGroup, record

    grp = ShardGroup(W.TATP, 50_000)
    grp.sync()
    grp.snapshot()
    d = Driver(W.TATP, 20_000, 50_000, zipf_theta=0.8)
    trace, done = record(d, grp, 25)
    assert sum(done) == d.stats()["txns"] > 0
    grp.sync()
    grp.restore()
    rp = Replay(trace, grp.msg)
    torch.cuda.synchronize()
    rp.run(grp, 0, 25)
    grp.sync()
    rp.check(0, 25)
    # replaying from the wrong state must be caught by check()
    rp.run(grp, 5, 25)
    grp.sync()
    with pytest.raises(AssertionError):
        rp.check(5, 25)
