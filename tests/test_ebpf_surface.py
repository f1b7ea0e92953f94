"""The eBPF-flavour protocol surface and the log drain (8f-4).

These request / reply codes exist only in the reference's eBPF servers: WARMUP_READ (smallbank/ebpf/shard_kern.c:585-667
+ shard_user.c:179-186), REJECT_LOCK_SAME_KEY (tatp/ebpf/lock_kern.c:289-298) and the back-pressure replies REJECT_* / RETRY.  These tests hold the engine to the
oracle's restatement; the restatement itself is pinned to the unmodified eBPF programs run under the emulator
(oracle/ref_harness/ebpf, tests/test_ebpf_golden.py)."""
import numpy as np
import pytest

import tracegen
from dint_amd import wire
from dint_amd.engine import refuse
from oracle import oracle as orc

W = wire.Workload


def test_refusal_codes_follow_the_ebpf_servers():
    T, S, B = wire.Tatp, wire.Store, wire.Sb
    m = np.zeros(14, wire.TATP_MSG)
    m["type"] = [T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.COMMIT_BCK, T.COMMIT_LOG, T.INSERT_PRIM, T.INSERT_BCK,
                 T.DELETE_PRIM, T.DELETE_BCK, T.DELETE_LOG, 3, 99, T.READ]
    m["key"], m["ver"], m["val"] = np.arange(14), 7, 0xAB
    r = refuse(W.TATP, m)
    assert r["type"].tolist() == [T.REJECT_READ, T.REJECT_LOCK, T.ABORT, T.REJECT_COMMIT, T.REJECT_COMMIT, T.COMMIT_LOG,
                                  T.REJECT_COMMIT, T.REJECT_COMMIT, T.REJECT_COMMIT, T.REJECT_COMMIT, T.DELETE_LOG, 3, 99,
                                  T.REJECT_READ]
    r2 = r.copy()
    r2["type"] = m["type"]
    assert r2.tobytes() == m.tobytes()  # nothing but the type byte changes
    s = np.zeros(4, wire.STORE_MSG)
    s["type"] = [S.READ, S.SET, S.INSERT, 7]
    assert refuse(W.STORE, s)["type"].tolist() == [S.REJECT_READ, S.REJECT_SET, S.REJECT_INSERT, 7]
    b = np.zeros(9, wire.SB_MSG)
    b["type"] = [0, 1, 2, 3, 4, 5, 6, 17, 9]
    assert refuse(W.SMALLBANK, b)["type"].tolist() == [B.RETRY] * 6 + [6, B.RETRY, 9]
    p = np.zeros(2, wire.TPL_MSG)
    p["action"] = [0, 1]
    assert refuse(W.TPL, p)["action"].tolist() == [wire.Tpl.RETRY] * 2
    f = np.zeros(4, wire.FASST_MSG)
    f["type"] = [0, 1, 2, 3]
    assert refuse(W.FASST, f)["type"].tolist() == [0, wire.Fasst.REJECT_LOCK, 2, 3]


def test_refusal_codes_match_the_unmodified_ebpf_programs_under_contention():
    """tests/golden/ebpf_backpressure.npz (make_golden_ebpf.py backpressure): the reference's own XDP programs, run under the
    emulator while it holds the spin lock of every cache entry / lock unit they look up -- the failed CAS of
    tatp/ebpf/shard_kern.c:173-178,371-376, store/ebpf/store_kern.c:57-66, lock_2pl/ebpf/ls_kern.c:59-64,
    smallbank/ebpf/shard_kern.c:122-152.  Where they answer a back-pressure code, dint_refuse (and through
    tests/test_gpu_route.py its device twin rt_refuse) must produce the same BYTES; a request they answer as always
    (tatp ACQUIRE_LOCK / ABORT and the log appends touch no spin lock) is one dint_refuse leaves in the batch -- but for
    tatp ACQUIRE_LOCK, which it refuses with the answer of a taken LOCK WORD, REJECT_LOCK (shard_kern.c:289-293): the one
    "not now" of that request type, pinned by every serial fixture."""
    import json
    import os

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ebpf_backpressure.npz"))
    meta = json.loads(str(z["meta"]))
    T = wire.Tatp
    back = {"lock_2pl": {wire.Tpl.RETRY}, "store": {wire.Store.REJECT_READ, wire.Store.REJECT_SET, wire.Store.REJECT_INSERT},
            "tatp": {T.REJECT_READ, T.REJECT_COMMIT}, "smallbank": {wire.Sb.RETRY}}
    for wl, dt, w in (("lock_2pl", wire.TPL_MSG, W.TPL), ("store", wire.STORE_MSG, W.STORE), ("tatp", wire.TATP_MSG, W.TATP),
                      ("smallbank", wire.SB_MSG, W.SMALLBANK)):
        req = np.frombuffer(z[wl + "_req"].tobytes(), dt)
        held = np.frombuffer(z[wl + "_rep"].tobytes(), dt)
        free = np.frombuffer(z[wl + "_free"].tobytes(), dt)
        f = "action" if wl == "lock_2pl" else "type"
        got = refuse(w, req)
        is_back = np.isin(held[f], list(back[wl]))
        assert is_back.sum() >= 48 and meta[wl]["n"] == len(req)
        assert got[is_back].tobytes() == held[is_back].tobytes(), wl      # the refusal, byte for byte
        assert (free[f][is_back] != held[f][is_back]).all()                 # ... which a serial replay never produces
        rest = ~is_back
        if wl == "tatp":
            acq = req["type"] == T.ACQUIRE_LOCK
            assert (got["type"][acq] == T.REJECT_LOCK).all() and (held["type"][acq] == T.GRANT_LOCK).all()
            rest &= ~acq
        assert got[rest].tobytes() == req[rest].tobytes(), wl               # not refusable: stays in the batch
        assert held[rest].tobytes() == free[rest].tobytes(), wl             # ... the eBPF program answers it as always


def test_oracle_same_key_and_warmup_semantics():
    T = wire.Tatp
    o = orc.TatpOracle(300, log_entries=1000)
    o.same_key_mode()
    m = np.zeros(5, wire.TATP_MSG)
    m["type"] = [T.ACQUIRE_LOCK, T.ACQUIRE_LOCK, T.ABORT, T.ACQUIRE_LOCK, T.ACQUIRE_LOCK]
    m["key"] = [5, 5, 5, 5, 5]
    assert o.replay(m)["type"].tolist() == [T.GRANT_LOCK, 28, T.ABORT_ACK, T.GRANT_LOCK, 28]
    s = orc.SmallbankOracle(2000, log_entries=1000)
    w = np.zeros(2, wire.SB_MSG)
    w["type"], w["key"], w["ver"] = 17, [3, 999_999], 0x55
    r = s.replay(w)
    assert r["type"].tolist() == [18, 18] and r["ver"].tolist() == [0, 0x55] and r["val"][0, 0] == 97 and r["val"][1, 0] == 0


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("rounds", [0, 1])
def test_smallbank_warmup_read_vs_oracle(rounds):
    from dint_amd.engine import Engine

    n = 60_000
    req = tracegen.sb_random(n, seed=9, n_acct_touch=30)
    rng = np.random.default_rng(1)
    w = rng.random(n) < 0.15
    req["type"][w] = 17  # WARMUP_READ mixed into lock / commit traffic on the same hot accounts
    req["key"][rng.random(n) < 0.01] = 5_000_000  # ... and on rows that do not exist
    e = Engine(W.SMALLBANK, n_rows=2000, log_entries=100_000, flags=rounds)
    e.populate(2000)
    o = orc.SmallbankOracle(2000, log_entries=100_000)
    assert e.submit(req).tobytes() == o.replay(req).tobytes()
    for t in range(2):
        ex, sh = e.read_locks(t)
        assert (ex == o.num_ex(t)).all() and (sh == o.num_sh(t)).all()
        assert all((x == y).all() for x, y in zip(e.dump_rows(t), o.dump(t)))


@pytest.mark.gpu
def test_tatp_reject_lock_same_key_vs_oracle():
    from dint_amd.engine import Engine

    o = orc.TatpOracle(300, log_entries=100_000)
    req = tracegen.tatp_random(50_000, [o.dump(t)[0] for t in range(5)], seed=12, n_sub_touch=6)  # few rows: many lock conflicts
    o.same_key_mode()
    e = Engine(W.TATP, n_rows=300, log_entries=100_000, flags=4)  # DINT_FLAG_LOCK_SAME_KEY
    e.populate(300)
    got, want = e.submit(req), o.replay(req)
    assert got.tobytes() == want.tobytes()
    assert (want["type"] == 28).sum() > 100 and (want["type"] == 8).sum() > 100
    # without the flag the same requests get plain REJECT_LOCK and nothing else differs
    e2 = Engine(W.TATP, n_rows=300, log_entries=100_000)
    e2.populate(300)
    plain = e2.submit(req)
    w2 = want.copy()
    w2["type"][w2["type"] == 28] = 8
    assert plain.tobytes() == w2.tobytes()


def _committed_writes(o, n, seed):
    """n committed single-row transactions as the client sends them to ONE server: the log record, then the primary op
    (tatp/caladan/client_udp_shard.cc:486-570), over a few dozen subscribers: updates, inserts and deletes"""
    T = wire.Tatp
    rng = np.random.default_rng(seed)
    live = [set(int(k) for k in o.dump(t)[0]) for t in range(5)]
    pools = [sorted(live[t] | {k + (1 << 44) for k in list(live[t])[:50]}) for t in range(5)]  # + keys that do not exist yet
    m = np.zeros(2 * n, wire.TATP_MSG)
    for i in range(n):
        t = int(rng.integers(0, 5))
        key = pools[t][int(rng.integers(0, len(pools[t])))]
        val = rng.integers(0, 256, 40, dtype=np.uint8)
        if key in live[t]:
            if rng.random() < 0.25:
                ops = (T.DELETE_LOG, T.DELETE_PRIM)
                live[t].discard(key)
            else:
                ops = (T.COMMIT_LOG, T.COMMIT_PRIM)
        else:
            ops = (T.COMMIT_LOG, T.INSERT_PRIM)
            live[t].add(key)
        for j in (0, 1):
            r = m[2 * i + j]
            r["type"], r["table"], r["key"], r["val"], r["ver"] = ops[j], t, key, val, i
    return m


@pytest.mark.gpu
def test_log_drain_and_replica_rebuild():
    from dint_amd import recovery
    from dint_amd.engine import Engine

    cap = 4096  # a small ring: it wraps several times
    prim = Engine(W.TATP, n_rows=300, log_entries=cap)
    prim.populate(300)
    o = orc.TatpOracle(300, log_entries=1 << 20)
    drained = []
    for b in range(8):
        req = _committed_writes(o, 3000, seed=b)
        assert prim.submit(req).tobytes() == o.replay(req).tobytes()
        rec, lost = prim.log_drain()
        assert lost == 0 and len(rec) == 3000
        drained.append(rec)
    rec, lost = prim.log_drain()
    assert len(rec) == 0 and lost == 0
    allrec = np.concatenate(drained)
    # the drained stream = the oracle's unwrapped ring; a DELETE_LOG record copies no value (server_shard.cc:196-207), so
    # its val bytes are whatever the ring slot held before -- which depends on the ring size: masked
    want = np.frombuffer(o.ring[:o.tail].tobytes(), wire.LOG_REC).copy()
    got = allrec.copy()
    assert (got["is_del"] == want["is_del"]).all() and got["is_del"].sum() > 100
    got["val"][got["is_del"] != 0] = 0
    want["val"][want["is_del"] != 0] = 0
    assert got.tobytes() == want.tobytes()
    # rebuild a replica from the log alone: rows and versions equal the primary's
    rep = Engine(W.TATP, n_rows=300, log_entries=cap)
    rep.populate(300)
    st = recovery.apply_log(rep, allrec)
    assert st["applied"] == len(allrec) and st["inserts"] > 0 and st["deletes"] > 0
    for t in range(5):
        a, b = prim.dump_rows(t), rep.dump_rows(t)
        ka, kb = np.argsort(a[0], kind="stable"), np.argsort(b[0], kind="stable")
        assert (a[0][ka] == b[0][kb]).all() and (a[1][ka] == b[1][kb]).all() and (a[2][ka] == b[2][kb]).all(), t
    assert rep.stats()["missing_keys"] == 0
    # an undrained ring that laps reports what it lost
    req = _committed_writes(o, 3000, seed=100)
    prim.submit(req)
    prim.submit(_committed_writes(o, 3000, seed=101))
    rec, lost = prim.log_drain()
    assert lost == 6000 - cap and len(rec) == cap
