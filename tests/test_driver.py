"""The closed-loop transaction drivers (restated reference clients) against three CPU oracle
shard servers: the reference client's own `assert`s serve as known-answer checks
(tatp/caladan/client_udp_shard.cc:189-195,398,421-422 ...), plus mix, commit-rate and determinism."""
import numpy as np
import pytest

from dint_amd import wire
from dint_amd.driver import Driver, run_epochs
from oracle import oracle as orc

T, S = wire.Tatp, wire.Sb


class OracleServer:
    def __init__(self, o):
        self.o = o

    def submit(self, req):
        return self.o.replay(req)


def tatp_servers(n_sub):
    return [OracleServer(orc.TatpOracle(n_sub, log_entries=100_000)) for _ in range(3)]


@pytest.mark.parametrize("zipf", [None, 0.8])
def test_tatp_driver_against_three_oracle_shards(zipf):
    n_sub, W, E = 20_000, 3000, 60
    srv = tatp_servers(n_sub)
    d = Driver(wire.Workload.TATP, W, n_sub, zipf_theta=zipf)
    trace = run_epochs(d, srv, E, record=True)
    st = d.stats()
    assert all(s.o.errors == 0 for s in srv)  # no unknown type, no commit/delete of a missing row
    assert st["epochs"] == E and st["txns"] > W * 10
    # transaction mix 35/35/10/2/14/2/2 (tatp.h:57-63); by_type is indexed by TxnType (tatp.h:45-53)
    # and counts FINISHED transactions, so long transactions are under-represented in a short run
    frac = np.array(st["by_type"][:7]) / st["txns"]
    assert 0.3 < frac[0] < 0.42 and 0.3 < frac[2] < 0.42 and 0.05 < frac[1] < 0.14
    # GET_SUBSCRIBER_DATA always commits (client assert :189); GET_ACCESS_DATA succeeds when the row exists (62.5%)
    assert st["committed_by_type"][0] == st["by_type"][0]
    ga = st["committed_by_type"][2] / st["by_type"][2]
    assert 0.5 < ga < 0.75
    # every reply type is one the client expects for its request (the reference's asserts)
    ok = {T.READ: {T.GRANT_READ, T.NOT_EXIST}, T.ACQUIRE_LOCK: {T.GRANT_LOCK, T.REJECT_LOCK}, T.ABORT: {T.ABORT_ACK},
          T.COMMIT_LOG: {T.COMMIT_LOG_ACK}, T.COMMIT_BCK: {T.COMMIT_BCK_ACK}, T.COMMIT_PRIM: {T.COMMIT_PRIM_ACK},
          T.INSERT_BCK: {T.INSERT_BCK_ACK}, T.INSERT_PRIM: {T.INSERT_PRIM_ACK}, T.DELETE_LOG: {T.DELETE_LOG_ACK},
          T.DELETE_BCK: {T.DELETE_BCK_ACK}, T.DELETE_PRIM: {T.DELETE_PRIM_ACK}}
    seen = set()
    for req, rep in trace:
        for s in range(3):
            for rt, pt in set(zip(req[s]["type"].tolist(), rep[s]["type"].tolist())):
                assert pt in ok[rt], (rt, pt)
                seen.add(rt)
            # routing: reads, locks and primary ops go to key % 3 (client_udp_shard.cc:187)
            prim = np.isin(req[s]["type"], [T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.INSERT_PRIM, T.DELETE_PRIM])
            assert (req[s]["key"][prim] % 3 == s).all()
            bck = np.isin(req[s]["type"], [T.COMMIT_BCK, T.INSERT_BCK, T.DELETE_BCK])
            assert (req[s]["key"][bck] % 3 != s).all()
    assert {T.READ, T.ACQUIRE_LOCK, T.COMMIT_LOG, T.COMMIT_BCK, T.COMMIT_PRIM, T.INSERT_PRIM, T.DELETE_PRIM} <= seen
    # log rings of the three shards hold the same records (every log goes to all three)
    tails = [s.o.tail for s in srv]
    assert len(set(tails)) == 1 and tails[0] > 0
    assert all((srv[0].o.ring[:tails[0]] == srv[s].o.ring[:tails[0]]).all() for s in (1, 2))
    # determinism
    srv2 = tatp_servers(n_sub)
    d2 = Driver(wire.Workload.TATP, W, n_sub, zipf_theta=zipf)
    trace2 = run_epochs(d2, srv2, E, record=True)
    assert all(a[0][s].tobytes() == b[0][s].tobytes() for a, b in zip(trace, trace2) for s in range(3))
    assert d2.stats() == st


def test_tatp_driver_conflicts_abort():
    """Few subscribers, many clients: lock rejects and validation failures must abort cleanly
    (every granted lock is released: after the clients drain, no lock is left held)."""
    n_sub, W = 50, 2000
    srv = tatp_servers(n_sub)
    d = Driver(wire.Workload.TATP, W, n_sub)
    run_epochs(d, srv, 80)
    st = d.stats()
    assert all(s.o.errors == 0 for s in srv)
    upd = st["by_type"][3] + st["by_type"][4]
    assert st["committed_by_type"][3] + st["committed_by_type"][4] < upd  # some update transactions aborted


def test_smallbank_driver_against_three_oracle_shards():
    n_acct, W, E = 600_000, 1000, 60  # 24k hot accounts (4%, smallbank.h:17-18) for 1000 clients
    srv = [OracleServer(orc.SmallbankOracle(n_acct, log_entries=100_000)) for _ in range(3)]
    d = Driver(wire.Workload.SMALLBANK, W, n_acct)
    trace = run_epochs(d, srv, E, record=True)
    st = d.stats()
    assert all(s.o.errors == 0 for s in srv)
    frac = np.array(st["by_type"][:6]) / st["txns"]
    assert st["txns"] > W * 5 and (frac > 0.08).all() and frac[3] > frac[0]  # SendPayment is the 25% one
    assert 0.7 < st["committed"] / st["txns"] <= 1.0
    ok = {S.ACQUIRE_SHARED: {S.GRANT_SHARED, S.REJECT_SHARED}, S.ACQUIRE_EXCLUSIVE: {S.GRANT_EXCLUSIVE, S.REJECT_EXCLUSIVE},
          S.RELEASE_SHARED: {S.RELEASE_SHARED_ACK}, S.RELEASE_EXCLUSIVE: {S.RELEASE_EXCLUSIVE_ACK},
          S.COMMIT_PRIM: {S.COMMIT_PRIM_ACK}, S.COMMIT_BCK: {S.COMMIT_BCK_ACK}, S.COMMIT_LOG: {S.COMMIT_LOG_ACK}}
    for req, rep in trace:
        for s in range(3):
            for rt, pt in set(zip(req[s]["type"].tolist(), rep[s]["type"].tolist())):
                assert pt in ok[rt], (rt, pt)
    # 2PL bookkeeping: counters never wrap (a release without a matching grant would make them huge)
    for s in srv:
        for t in range(2):
            assert s.o.num_ex(t).max() <= 1 and s.o.num_sh(t).max() < W
    # money is conserved up to the deposits the committed transactions made: spot-check the magic bytes
    for s in srv:
        k, v, vals = s.o.dump(0)
        assert (vals[:, 0] == 97).all()
