"""GPU parity: store / tatp / smallbank shard servers through the C ABI vs the CPU oracle and the
golden fixtures recorded from the unmodified reference udp/ servers.  Bit-exact: reply streams,
final rows (bucket order, chain order), lock words and log rings."""
import json
import os

import numpy as np
import pytest

import tracegen
from dint_amd import wire
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
W = wire.Workload


def _engine(*a, **k):
    from dint_amd.engine import Engine

    return Engine(*a, **k)


def _golden(name, dtype):
    z = np.load(os.path.join(G, name + ".npz"))
    return z, json.loads(str(z["meta"])), np.frombuffer(z["req"].tobytes(), dtype), np.frombuffer(z["rep"].tobytes(), dtype)


def _same_rows(a, b):
    return all(x.shape == y.shape and (x == y).all() for x, y in zip(a, b))


# ---------------------------------------------------------------- store
def test_store_golden_reference_sizes():
    z, meta, req, rep = _golden("store", wire.STORE_MSG)
    eng = _engine(W.STORE, n_rows=meta["n_sub"])
    assert eng.hash_size(0) == meta["hash_size"]
    eng.populate(meta["touch"])
    got = eng.submit(req)
    a = tracegen.mask_populate_garbage("store", got)
    b = tracegen.mask_populate_garbage("store", rep)
    assert a.tobytes() == b.tobytes()


def test_store_kat3():
    """SURVEY.md 8(c) KAT-3, recorded from the unmodified store/udp/server.cc."""
    S = wire.Store
    eng = _engine(W.STORE, n_rows=1000)
    eng.populate(1)
    key = 1 << 32  # s_id 0, sf_type 1, start_time 0

    def one(t, k, v0=None):
        m = np.zeros(1, wire.STORE_MSG)
        m["type"], m["key"], m["ver"] = t, k, 0x11223344
        m["val"] = 0xEE
        if v0 is not None:
            m["val"][0, 0] = v0
        return eng.submit(m)[0]

    r = one(S.READ, key)
    assert r["type"] == 3 and r["val"][0] == 21 and r["val"][1] == 0x5A and r["ver"] == 0
    r = one(S.SET, key, 7)
    assert r["type"] == 5 and r["ver"] == 0x11223344 and r["val"][0] == 7 and r["val"][1] == 0xEE
    r = one(S.READ, key)
    assert r["val"][0] == 7 and r["val"][2] == 0xEE and r["ver"] == 1
    assert one(S.SET, key, 9)["type"] == 5
    r = one(S.READ, key)
    assert r["val"][0] == 9 and r["ver"] == 2
    r = one(S.READ, 0xDEAD << 48)
    assert r["type"] == 7 and r["ver"] == 0x11223344 and (r["val"] == 0xEE).all()
    assert one(S.SET, 0xDEAD << 48)["type"] == 7


@pytest.mark.parametrize("n,n_sub,touch,p_set,p_ins", [
    (1, 1000, 50, 0.4, 0.0), (64, 1000, 3, 0.5, 0.0), (5000, 1000, 50, 0.4, 0.05), (65536, 50_000, 2000, 0.3, 0.02),
    (150_000, 2000, 40, 0.5, 0.1), (65536, 1000, 1, 0.6, 0.0),
])
def test_store_vs_oracle(n, n_sub, touch, p_set, p_ins):
    hs = n_sub * 18 // 4
    req = tracegen.store_random(n, seed=n + touch, n_sub_touch=touch, p_set=p_set, p_insert=p_ins)
    eng = _engine(W.STORE, n_rows=n_sub)
    eng.populate(min(touch * 2, n_sub))
    o = orc.StoreOracle(hs, min(touch * 2, n_sub))
    assert _same_rows(eng.dump_rows(0), o.dump())
    got, want = eng.submit(req), o.replay(req)
    assert got.tobytes() == want.tobytes()
    assert _same_rows(eng.dump_rows(0), o.dump())
    st = eng.stats()
    assert st["bad_requests"] == o.errors == 0 and st["pool_exhausted"] == 0


def test_store_bad_types_batch_split_and_device_path():
    import torch

    req = tracegen.store_random(40_000, seed=77, n_sub_touch=30, p_set=0.5, p_insert=0.05)
    req["type"][::13] = 99
    want_o = orc.StoreOracle(4500, 60)
    want = want_o.replay(req)
    for bs in (64, 4096, 65536):
        eng = _engine(W.STORE, n_rows=1000)
        eng.populate(60)
        got = np.concatenate([eng.submit(req[i:i + bs]) for i in range(0, len(req), bs)])
        assert got.tobytes() == want.tobytes(), bs
        assert eng.stats()["bad_requests"] == want_o.errors
    eng = _engine(W.STORE, n_rows=1000)
    eng.populate(60)
    d = torch.from_numpy(np.frombuffer(req.tobytes(), np.uint8).copy()).cuda()
    out = torch.empty_like(d)
    eng.submit_device(d, len(req), out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == want.tobytes()
    assert d.cpu().numpy().tobytes() == req.tobytes()  # out-of-place: requests untouched


def test_store_full_population_matches_oracle_dump():
    """Every row of a 100k-subscriber store (1.2M rows): same rows, same bucket/chain order."""
    n = 100_000
    eng = _engine(W.STORE, n_rows=n)
    eng.populate(n)
    o = orc.StoreOracle(n * 18 // 4, n)
    a, b = eng.dump_rows(0), o.dump()
    assert len(a[0]) == 12 * n and _same_rows(a, b)
    assert eng.stats()["pool_exhausted"] == 0


# ---------------------------------------------------------------- tatp
def _tatp_locks(eng, o):
    for t in range(5):
        lk, _ = eng.read_locks(t)
        assert (lk == o.locks(t)).all(), t


def test_tatp_golden_reference_sizes():
    z, meta, req, rep = _golden("tatp", wire.TATP_MSG)
    eng = _engine(W.TATP, n_rows=meta["n_sub"])
    eng.populate(meta["touch"])
    got = eng.submit(req)
    a = tracegen.mask_populate_garbage("tatp", got)
    b = tracegen.mask_populate_garbage("tatp", rep)
    assert a.tobytes() == b.tobytes()
    st = eng.stats()
    assert st["bad_requests"] == 0 and st["missing_keys"] == 0
    tail_dump = z["dump_tail"].tobytes()
    off = 0
    for t in range(5):
        cnt = int(np.frombuffer(tail_dump, "<u4", 1, off)[0]); off += 4
        held = np.frombuffer(tail_dump, "<u4", cnt, off); off += 4 * cnt
        lk, _ = eng.read_locks(t)
        assert (np.nonzero(lk)[0] == held).all()
    tail, n = np.frombuffer(tail_dump, "<u4", 2, off); off += 8
    recs = np.frombuffer(tail_dump, "u1", n * 64, off).reshape(n, 64)
    ring, t_eng = eng.read_log(int(n))
    ring = np.frombuffer(ring.tobytes(), "u1").reshape(n, 64)
    assert t_eng == tail
    is_del = recs[:, 52] == 1
    assert (ring[:, :8] == recs[:, :8]).all() and (ring[:, 48:54] == recs[:, 48:54]).all()
    assert (ring[~is_del, 8:48] == recs[~is_del, 8:48]).all()


@pytest.mark.parametrize("n,n_sub,touch,well", [
    (1, 2000, 40, True), (4096, 2000, 40, True), (65536, 2000, 40, True), (70_000, 100_000, 3000, True),
    (30_000, 2000, 6, False), (100_000, 2000, 2, True),
])
def test_tatp_vs_oracle(n, n_sub, touch, well):
    o = orc.TatpOracle(n_sub, log_entries=100_000, populate_n=touch)
    existing = [o.dump(t)[0] for t in range(5)]
    req = tracegen.tatp_random(n, existing, seed=n + touch, n_sub_touch=touch, well_formed=well)
    if not well:
        req["type"][7::41] = 3      # kCommit is never handled by the reference servers
        req["table"][11::53] = 9    # out-of-range table
    eng = _engine(W.TATP, n_rows=n_sub, log_entries=100_000)
    eng.populate(touch)
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t)), t
    got, want = eng.submit(req), o.replay(req)
    assert got.tobytes() == want.tobytes()
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t)), t
    _tatp_locks(eng, o)
    ring, tail = eng.read_log(100_000)
    assert tail == o.tail
    assert (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64) == o.ring).all()
    st = eng.stats()
    assert st["bad_requests"] + st["missing_keys"] == o.errors and st["pool_exhausted"] == 0
    if well:
        assert o.errors == 0


def test_tatp_log_ring_wrap_and_small_ring():
    """A ring smaller than a micro-batch: passes are clamped to the ring size, so DELETE_LOG records
    (which keep the val bytes of the slot they overwrite) still match the serial reference."""
    o = orc.TatpOracle(2000, log_entries=777, populate_n=20)
    existing = [o.dump(t)[0] for t in range(5)]
    req = tracegen.tatp_random(30_000, existing, seed=5, n_sub_touch=20)
    eng = _engine(W.TATP, n_rows=2000, log_entries=777)
    eng.populate(20)
    assert eng.submit(req).tobytes() == o.replay(req).tobytes()
    ring, tail = eng.read_log(777)
    assert tail == o.tail and (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64) == o.ring).all()


def test_tatp_full_population_1m_subscribers():
    """BASELINE.json configs[3] size: 1M subscribers -- every row of all five tables equals the oracle's
    restatement of tatp/udp/tatp.h:283-412, in bucket and chain order; then a 3x64k trace over it."""
    n_sub = 1_000_000
    eng = _engine(W.TATP, n_rows=n_sub)
    eng.populate(n_sub)
    o = orc.TatpOracle(n_sub)
    existing = []
    for t in range(5):
        a, b = eng.dump_rows(t), o.dump(t)
        assert _same_rows(a, b), t
        existing.append(b[0][:4000])
    assert eng.stats()["pool_exhausted"] == 0
    rng = np.random.default_rng(1)
    n = 3 * 65536
    req = np.zeros(n, wire.TATP_MSG)
    T = wire.Tatp
    tb = rng.integers(0, 5, n)
    for t in range(5):
        sel = tb == t
        pool = o.dump(t)[0]
        req["key"][sel] = pool[rng.integers(0, len(pool), int(sel.sum()))]
    req["table"] = tb
    req["type"] = rng.choice([T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.COMMIT_BCK, T.COMMIT_LOG], n,
                             p=[0.5, 0.15, 0.05, 0.1, 0.1, 0.1])
    req["val"] = rng.integers(0, 256, (n, 40), dtype=np.uint8)
    req["ver"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype("<u4")
    assert eng.submit(req).tobytes() == o.replay(req).tobytes()
    _tatp_locks(eng, o)


# ---------------------------------------------------------------- smallbank
def _sb_state(eng, o):
    for t in range(2):
        ex, sh = eng.read_locks(t)
        assert (ex == o.num_ex(t)).all() and (sh == o.num_sh(t)).all()
        assert _same_rows(eng.dump_rows(t), o.dump(t))


def test_smallbank_golden_reference_sizes():
    z, meta, req, rep = _golden("smallbank", wire.SB_MSG)
    eng = _engine(W.SMALLBANK, n_rows=meta["n_acct"])
    eng.populate(meta["touch"])
    assert eng.submit(req).tobytes() == rep.tobytes()
    tail_dump = z["dump_tail"].tobytes()
    off = 0
    for t in range(2):
        cnt = int(np.frombuffer(tail_dump, "<u4", 1, off)[0]); off += 4
        d = np.frombuffer(tail_dump, "<u4", cnt * 3, off).reshape(cnt, 3); off += 12 * cnt
        ex, sh = eng.read_locks(t)
        nz = np.nonzero(ex | sh)[0]
        assert (d[:, 0] == nz).all() and (d[:, 1] == ex[nz]).all() and (d[:, 2] == sh[nz]).all()


@pytest.mark.parametrize("n,n_acct,touch", [(1, 10_000, 40), (5000, 10_000, 40), (65536, 10_000, 3),
                                            (200_000, 1_000_000, 5000), (65536, 10_000, 1)])
def test_smallbank_vs_oracle(n, n_acct, touch):
    req = tracegen.sb_random(n, seed=n + touch, n_acct_touch=touch)
    req["type"][9::31] = 17  # WARMUP_READ: the udp server has no handler (Appendix B.6) -> counted, echoed
    if n > 100:
        req["key"][5::97] = 10**9  # lock + read of a missing account: the reference panics; counted
    eng = _engine(W.SMALLBANK, n_rows=n_acct, log_entries=70_000)
    eng.populate(touch)
    o = orc.SmallbankOracle(n_acct, log_entries=70_000, populate_n=touch)
    got, want = eng.submit(req), o.replay(req)
    assert got.tobytes() == want.tobytes()
    _sb_state(eng, o)
    ring, tail = eng.read_log(70_000)
    assert tail == o.tail and (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64) == o.ring).all()
    st = eng.stats()
    assert st["bad_requests"] + st["missing_keys"] == o.errors


# ---------------------------------------------------------------- sharding on one GPU
@pytest.mark.parametrize("wl", ["store", "tatp", "smallbank"])
def test_two_shards_on_one_gpu_equal_one_server(wl):
    """Two engines (shard 0/2 and 1/2) fed the requests they are home to, in order, answer exactly
    like one unsharded server; requests sent to the wrong shard are left untouched and counted."""
    import torch

    if wl == "store":
        req = tracegen.store_random(40_000, seed=3, n_sub_touch=200, p_set=0.4, p_insert=0.05)
        mk = lambda **k: _engine(W.STORE, n_rows=5000, **k)
        o = orc.StoreOracle(5000 * 18 // 4, 400); pop = 400
    elif wl == "tatp":
        o = orc.TatpOracle(5000, populate_n=300); pop = 300
        req = tracegen.tatp_random(40_000, [o.dump(t)[0] for t in range(5)], seed=4, n_sub_touch=300)
        T = wire.Tatp
        req = req[(req["type"] != T.COMMIT_LOG) & (req["type"] != T.DELETE_LOG)]  # logs are per-shard rings
        mk = lambda **k: _engine(W.TATP, n_rows=5000, **k)
    else:
        req = tracegen.sb_random(40_000, seed=5, n_acct_touch=300)
        req = req[req["type"] != wire.Sb.COMMIT_LOG]
        mk = lambda **k: _engine(W.SMALLBANK, n_rows=5000, **k)
        o = orc.SmallbankOracle(5000, populate_n=300); pop = 300
    want = o.replay(req)
    shards = [mk(shard_index=i, shard_count=2) for i in range(2)]
    for s in shards:
        s.populate(pop)
    d = torch.from_numpy(np.frombuffer(req.tobytes(), np.uint8).copy()).cuda()
    home = torch.empty(len(req), dtype=torch.uint8, device="cuda")
    shards[0].home_shard(d, len(req), home)
    torch.cuda.synchronize()
    home = home.cpu().numpy()
    assert set(np.unique(home)) <= {0, 1} and 0.3 < home.mean() < 0.7
    got = req.copy()
    for i, s in enumerate(shards):
        sel = home == i
        got[sel] = s.submit(req[sel])
        assert s.stats()["foreign_requests"] == 0
    assert got.tobytes() == want.tobytes()
    # wrong shard: untouched + counted
    wrong = shards[0].submit(req[home == 1][:100])
    assert wrong.tobytes() == req[home == 1][:100].tobytes()
    assert shards[0].stats()["foreign_requests"] == 100


# ---------------------------------------------------------------- closed-form vs request-by-request resolution
@pytest.mark.parametrize("flags", [0, 1])  # 1 = DINT_FLAG_KV_ROUNDS
def test_hot_key_paths_agree_with_oracle(flags):
    """Zipf-like traffic puts hundreds of requests of a pass on one key.  Both resolution strategies of the
    resolve kernel (closed-form same-key groups; one round per request) must give the serial answer."""
    # store: 12 keys, 70% SET
    req = tracegen.store_random(30_000, seed=31, n_sub_touch=1, p_set=0.7, p_missing=0.05)
    eng = _engine(W.STORE, n_rows=1000, flags=flags)
    eng.populate(2)
    o = orc.StoreOracle(4500, 2)
    assert eng.submit(req).tobytes() == o.replay(req).tobytes()
    assert _same_rows(eng.dump_rows(0), o.dump())
    # tatp: 3 subscribers, every request type incl. insert/delete (which force the round path for their bucket)
    o = orc.TatpOracle(2000, log_entries=100_000, populate_n=3)
    req = tracegen.tatp_random(40_000, [o.dump(t)[0] for t in range(5)], seed=32, n_sub_touch=3)
    eng = _engine(W.TATP, n_rows=2000, log_entries=100_000, flags=flags)
    eng.populate(3)
    assert eng.submit(req).tobytes() == o.replay(req).tobytes()
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t))
    _tatp_locks(eng, o)
    # smallbank: 2 accounts
    req = tracegen.sb_random(40_000, seed=33, n_acct_touch=2)
    eng = _engine(W.SMALLBANK, n_rows=10_000, log_entries=100_000, flags=flags)
    eng.populate(2)
    o = orc.SmallbankOracle(10_000, log_entries=100_000, populate_n=2)
    assert eng.submit(req).tobytes() == o.replay(req).tobytes()
    _sb_state(eng, o)


def test_store_baseline_config_16m_keys_95_5():
    """BASELINE.json configs[2]: store KV, 16M keys (1.4M subscribers x 12 rows = 16.8M), 95/5 read/write.
    Full population equals the oracle's row for row; 4 x 64k requests with tatp_nurand-like skew, bit-exact."""
    n_sub = 1_400_000
    eng = _engine(W.STORE, n_rows=n_sub)
    eng.populate(n_sub)
    o = orc.StoreOracle(n_sub * 18 // 4, n_sub)
    rng = np.random.default_rng(2)
    n = 4 * 65536
    req = np.zeros(n, wire.STORE_MSG)
    s_id = (rng.integers(0, n_sub, n) | (rng.integers(0, 1 << 20, n) & 0xFFFFF)) % n_sub  # NURand(A = 2^20 - 1)
    req["key"] = tracegen.store_key(s_id, rng.integers(1, 5, n), rng.integers(0, 3, n) * 8)
    req["type"] = np.where(rng.random(n) < 0.05, wire.Store.SET, wire.Store.READ)
    req["val"] = rng.integers(0, 256, (n, 40), dtype=np.uint8)
    got, want = eng.submit(req), o.replay(req)
    assert got.tobytes() == want.tobytes()
    assert (got["type"] == wire.Store.GRANT_READ).sum() > 0.9 * n
    a, b = eng.dump_rows(0), o.dump()
    assert len(a[0]) == 12 * n_sub and _same_rows(a, b)
    assert eng.stats()["pool_exhausted"] == 0


# ---------------------------------------------------------------- pass size
@pytest.mark.parametrize("max_pass", [0, 65536, 5000, 777])
def test_pass_size_does_not_change_the_answer(max_pass):
    """One submission = ceil(n / max_pass) kernel passes; the reply stream, rows, locks and log are those of the
    serial reference whatever the split (0 = the engine's largest pass, 2^20 requests)."""
    o = orc.TatpOracle(50_000, populate_n=700)
    req = tracegen.tatp_random(180_000, [o.dump(t)[0] for t in range(5)], seed=51, n_sub_touch=700)
    eng = _engine(W.TATP, n_rows=50_000, max_pass=max_pass)
    eng.populate(700)
    assert eng.submit(req).tobytes() == o.replay(req).tobytes()
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t)), t
    _tatp_locks(eng, o)
    ring, tail = eng.read_log(1_000_000)
    assert tail == o.tail and (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64)[:tail] == o.ring[:tail]).all()
    assert eng.stats()["batches"] == -(-180_000 // (max_pass or (1 << 20)))

    req = tracegen.sb_random(150_000, seed=52, n_acct_touch=300)
    eng = _engine(W.SMALLBANK, n_rows=100_000, max_pass=max_pass)
    eng.populate(300)
    o = orc.SmallbankOracle(100_000, populate_n=300)
    assert eng.submit(req).tobytes() == o.replay(req).tobytes()
    _sb_state(eng, o)


@pytest.mark.parametrize("wl,n,touch", [("store", 1_000_000, 20_000), ("store", 400_000, 1), ("tatp", 1_000_000, 30_000),
                                        ("tatp", 300_000, 2), ("smallbank", 1_000_000, 40_000), ("smallbank", 250_000, 1),
                                        ("store", 1_000_000, 250), ("smallbank", 1_000_000, 900)])
def test_one_pass_of_up_to_a_million_requests(wl, n, touch):
    """The largest passes (20-bit request index, 32768 bins), spread wide and piled on one or two hot keys (bins of
    10^5 records: hundreds of 512-request windows cut along request-index buckets) -- and over a few thousand keys of
    ~300 requests each: thousands of big bins, so that every workgroup of k_kv_scan_place runs its scan of the big-bin
    list in several trips."""
    if wl == "store":
        req = tracegen.store_random(n, seed=n + touch, n_sub_touch=touch, p_set=0.4, p_insert=0.02)
        eng = _engine(W.STORE, n_rows=100_000)
        eng.populate(min(2 * touch, 100_000))
        o = orc.StoreOracle(100_000 * 18 // 4, min(2 * touch, 100_000))
        assert eng.submit(req).tobytes() == o.replay(req).tobytes()
        assert _same_rows(eng.dump_rows(0), o.dump())
    elif wl == "tatp":
        o = orc.TatpOracle(100_000, populate_n=touch)
        req = tracegen.tatp_random(n, [o.dump(t)[0] for t in range(5)], seed=n + touch, n_sub_touch=touch)
        eng = _engine(W.TATP, n_rows=100_000)
        eng.populate(touch)
        assert eng.submit(req).tobytes() == o.replay(req).tobytes()
        for t in range(5):
            assert _same_rows(eng.dump_rows(t), o.dump(t)), t
        _tatp_locks(eng, o)
        ring, tail = eng.read_log(1_000_000)
        assert tail == o.tail and (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64) == o.ring).all()
    else:
        req = tracegen.sb_random(n, seed=n + touch, n_acct_touch=touch)
        eng = _engine(W.SMALLBANK, n_rows=1_000_000)
        eng.populate(touch)
        o = orc.SmallbankOracle(1_000_000, populate_n=touch)
        assert eng.submit(req).tobytes() == o.replay(req).tobytes()
        _sb_state(eng, o)
    assert eng.stats()["batches"] == 1 and eng.stats()["pool_exhausted"] == 0


@pytest.mark.parametrize("n,hot", [(3000, False), (120_000, True)])
def test_tatp_keys_sharing_a_lock_byte(n, hot):
    """Subscribers whose keys fall into one bucket and one lock quadrant share a lock byte (tatp/udp/tatp.h:12-14,
    lock_hash % hash_size == bucket).  ACQUIRE / ABORT / COMMIT_PRIM streams on such keys interleave on that byte;
    the engine resolves them without serialising the bucket (the byte is a last-writer-wins register)."""
    import struct

    n_sub = 3000
    o = orc.TatpOracle(n_sub, populate_n=n_sub)
    hs = o.hash_size(0)
    groups = {}
    for s in range(n_sub):  # subscriber table: key = s_id
        h = orc.fasthash64(struct.pack("<Q", s))
        groups.setdefault((h % hs, (h % (4 * hs)) // hs), []).append(s)
    shared = [g for g in groups.values() if len(g) >= 2]
    assert len(shared) >= 20
    rng = np.random.default_rng(7)
    T = wire.Tatp
    keys = np.array([s for g in shared[:6] for s in g[:3]], dtype=np.uint64)
    if hot:   # nearly everything on the first group: its bin holds ~10^5 records
        pk = np.r_[np.full(len(shared[0][:3]), 0.9 / len(shared[0][:3])), np.full(len(keys) - len(shared[0][:3]), 0.1 / (len(keys) - len(shared[0][:3])))]
    else:
        pk = np.full(len(keys), 1.0 / len(keys))
    req = np.zeros(n, wire.TATP_MSG)
    req["key"] = rng.choice(keys, n, p=pk)
    req["table"] = 0
    req["type"] = rng.choice([T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.COMMIT_BCK], n, p=[0.3, 0.35, 0.15, 0.12, 0.08])
    req["val"] = rng.integers(0, 256, (n, 40), dtype=np.uint8)
    req["ver"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype("<u4")
    for flags in (0, 1):
        eng = _engine(W.TATP, n_rows=n_sub, flags=flags)
        eng.populate(n_sub)
        oo = orc.TatpOracle(n_sub, populate_n=n_sub)
        want = oo.replay(req)
        got = eng.submit(req)
        assert got.tobytes() == want.tobytes()
        assert (want["type"] == T.REJECT_LOCK).sum() > n // 20 and (want["type"] == T.GRANT_LOCK).sum() > n // 50
        assert _same_rows(eng.dump_rows(0), oo.dump(0))
        _tatp_locks(eng, oo)


# ---------------------------------------------------------------- dominant key of a big bin (hot rows)
def _hot_tatp(n, p_hot, mix, seed, hot_key=(0, 7), existing=None, n_noise_sub=2000):
    """n requests: a fraction p_hot on ONE row (table, key) with the simple op types in `mix` (type -> weight), the rest
    random well-formed traffic; returns the request array"""
    rng = np.random.default_rng(seed)
    req = tracegen.tatp_random(n, existing, seed=seed + 1, n_sub_touch=n_noise_sub)
    hot = rng.random(n) < p_hot
    types, w = zip(*mix.items())
    ty = rng.choice(types, n, p=np.array(w, float) / sum(w))
    req["type"][hot] = ty[hot]
    req["table"][hot] = hot_key[0]
    req["key"][hot] = hot_key[1]
    return req


@pytest.mark.parametrize("p_hot,mix,hot_key", [
    (0.6, {0: 70, 1: 10, 2: 4, 12: 8, 13: 8}, (0, 7)),           # reads + a few hundred lock ops / writers
    (0.9, {0: 99, 13: 1}, (0, 7)),                               # almost only reads
    (0.5, {0: 30, 1: 30, 2: 20, 12: 10, 13: 10}, (0, 7)),        # > 1024 ordering ops in a stretch: four per thread
    (0.9, {0: 20, 1: 30, 2: 20, 12: 15, 13: 15}, (0, 7)),        # > 2048 ordering ops in a stretch: general path
    (0.6, {0: 70, 1: 10, 2: 4, 12: 8, 13: 8}, (0, 5_000_000)),   # the hot row does not exist
    (0.6, {0: 70, 1: 10, 2: 4, 12: 8, 13: 8}, (4, 7 | (1 << 32))),  # a CALL_FORWARDING row: inserts / deletes around it
    # the hot CALL_FORWARDING row itself inserted (also when it exists: duplicate rows) and deleted: its runs go request
    # by request, reads between two row changes in one round, refused ACQUIREs behind the first in one round
    (0.3, {0: 85, 1: 10, 2: 2, 18: 1.5, 22: 1.5}, (4, 7 | (1 << 32))),
    (0.02, {0: 85, 1: 12, 2: 1, 18: 1, 22: 1}, (4, 7 | (1 << 32))),
])
def test_tatp_dominant_key_vs_oracle(p_hot, mix, hot_key):
    n_sub = 3000
    o = orc.TatpOracle(n_sub, log_entries=400_000)
    existing = [o.dump(t)[0] for t in range(5)]
    eng = _engine(W.TATP, n_rows=n_sub, log_entries=400_000)
    eng.populate(n_sub)
    for k, n in enumerate((6000, 40_000, 150_000)):  # one stretch ... several stretches per bin
        req = _hot_tatp(n, p_hot, mix, seed=10 * k + 3, hot_key=hot_key, existing=existing, n_noise_sub=n_sub)
        got, want = eng.submit(req), o.replay(req)
        assert got.tobytes() == want.tobytes(), (k, np.nonzero(np.frombuffer(got.tobytes(), "u1") != np.frombuffer(want.tobytes(), "u1"))[0][:5] // 55)
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t)), t
        lk, _ = eng.read_locks(t)
        assert (lk == o.locks(t)).all()
    st = eng.stats()
    assert st["bad_requests"] == 0


def test_store_dominant_key_vs_oracle():
    n_sub = 5000
    o = orc.StoreOracle(n_sub * 18 // 4, n_sub)
    eng = _engine(W.STORE, n_rows=n_sub)
    eng.populate(n_sub)
    rng = np.random.default_rng(4)
    for n in (5000, 120_000):
        req = tracegen.store_random(n, seed=n, n_sub_touch=n_sub, p_set=0.3, p_missing=0.05)
        hot = rng.random(n) < 0.7
        req["key"][hot] = tracegen.store_key(11, 2, 8)
        got, want = eng.submit(req), o.replay(req)
        assert got.tobytes() == want.tobytes()
    assert _same_rows(eng.dump_rows(0), o.dump())


@pytest.mark.parametrize("same_quadrant", [False, True])
def test_tatp_two_hot_rows_in_one_bucket(same_quadrant):
    """Two hot subscribers whose rows share a bucket land in one sub: the dominant-key path answers the hotter one, is run
    again on what is left and answers the second (other lock quadrant) -- or, when they share the lock byte, leaves the
    stretch to the general path.  Either way byte for byte the oracle's replies and state."""
    import struct

    n_sub = 3000
    o = orc.TatpOracle(n_sub, populate_n=n_sub, log_entries=400_000)
    hs = o.hash_size(0)
    by_bucket = {}
    for s_id in range(n_sub):
        h = orc.fasthash64(struct.pack("<Q", s_id))
        by_bucket.setdefault(h % hs, []).append((s_id, (h % (4 * hs)) // hs))
    pair = None
    for g in by_bucket.values():
        for a in g:
            for b in g:
                if a[0] < b[0] and (a[1] == b[1]) == same_quadrant:
                    pair = pair or (a[0], b[0])
    assert pair is not None
    existing = [o.dump(t)[0] for t in range(5)]
    eng = _engine(W.TATP, n_rows=n_sub, log_entries=400_000)
    eng.populate(n_sub)
    rng = np.random.default_rng(5)
    T = wire.Tatp
    for k, n in enumerate((9000, 60_000)):
        req = tracegen.tatp_random(n, existing, seed=31 + k, n_sub_touch=n_sub)
        u = rng.random(n)
        for key, lo, hi in ((pair[0], 0.0, 0.55), (pair[1], 0.55, 0.85)):
            hot = (u >= lo) & (u < hi)
            req["table"][hot] = 0
            req["key"][hot] = key
            req["type"][hot] = rng.choice([T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.COMMIT_BCK], int(hot.sum()), p=[0.7, 0.1, 0.04, 0.08, 0.08])
        got, want = eng.submit(req), o.replay(req)
        assert got.tobytes() == want.tobytes(), k
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t)), t
    _tatp_locks(eng, o)


# ---------------------------------------------------------------- a hot key in pieces, several workgroups at once (r05)
SPLIT_KNOBS = [{"DINT_KV_SPLIT_MIN": "65", "DINT_KV_SPLIT_TARGET": "16"},    # 16 pieces of a few dozen requests, never a solo item
               {"DINT_KV_SPLIT_MIN": "200", "DINT_KV_SPLIT_TARGET": "100"},
               {},                                                          # the defaults: every big sub, ~384 requests per piece
               {"DINT_KV_NO_SPLIT": "1"},                                   # r04: one workgroup per hot key
               {"DINT_KV_LATE_BIG": "1"},                                   # r05: what k_kv_hot leaves goes to k_kv_big, not k_kv_late
               {"DINT_KV_NO_FUSE": "1"}]                                    # r05's launches: k_kv_resolve -> k_kv_hot -> k_kv_late


@pytest.mark.parametrize("knobs", SPLIT_KNOBS, ids=lambda k: "+".join(f"{a[8:]}={b}" for a, b in k.items()) or "default")
@pytest.mark.parametrize("p_hot,mix,hot_key", [
    (0.6, {0: 70, 1: 10, 2: 4, 12: 8, 13: 8}, (0, 7)),           # reads, lock ops and writers in every piece
    (0.9, {0: 99, 13: 1}, (0, 7)),                               # almost only reads: a writer in one piece or another
    (0.7, {0: 20, 1: 30, 2: 20, 12: 15, 13: 15}, (0, 7)),        # mostly ordering ops
    (0.6, {0: 70, 1: 10, 2: 4, 12: 8, 13: 8}, (0, 5_000_000)),   # the hot row does not exist (NOT_EXIST, missing_keys)
    (0.6, {0: 70, 1: 10, 2: 4, 12: 8, 13: 8}, (4, 7 | (1 << 32))),  # a CALL_FORWARDING row: the remainder inserts / deletes around it
    (0.5, {0: 85, 1: 10, 2: 2, 18: 1.5, 22: 1.5}, (4, 7 | (1 << 32))),  # the hot row itself inserted / deleted: not in closed form, piece 0 takes the sub
])
def test_tatp_hot_key_in_pieces(p_hot, mix, hot_key, knobs, monkeypatch):
    """k_kv_resolve cuts a sub that is one key into pieces by request-index range and k_kv_big answers every piece with a
    workgroup of its own (kv_hot_item): version / value / lock byte seen by a request come from its piece and from one word
    per earlier piece.  Small thresholds force the path on small passes; every variant -- and r04's single workgroup -- must
    give the oracle's bytes, rows and lock words, pass after pass on one engine (the words are tagged per pass)."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    n_sub = 3000
    o = orc.TatpOracle(n_sub, log_entries=400_000)
    existing = [o.dump(t)[0] for t in range(5)]
    eng = _engine(W.TATP, n_rows=n_sub, log_entries=400_000)
    eng.populate(n_sub)
    miss0 = 0
    for k, n in enumerate((900, 3000, 500, 6000, 9000, 250, 2500, 14_000, 600)):  # (a few hundred hot requests: a solo item, one workgroup)
        req = _hot_tatp(n, p_hot, mix, seed=17 * k + 5, hot_key=hot_key, existing=existing, n_noise_sub=n_sub)
        got, want = eng.submit(req), o.replay(req)
        assert got.tobytes() == want.tobytes(), (k, np.nonzero(np.frombuffer(got.tobytes(), "u1") != np.frombuffer(want.tobytes(), "u1"))[0][:5] // 55)
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t)), t
        lk, _ = eng.read_locks(t)
        assert (lk == o.locks(t)).all()
    st = eng.stats()
    assert st["bad_requests"] == 0 and st["big_bin_requests"] > 0
    if hot_key == (0, 5_000_000):
        assert st["missing_keys"] > 100  # every COMMIT of the missing hot row is counted (tatp/udp/kvs.h:91)


@pytest.mark.parametrize("knobs", SPLIT_KNOBS[:3] + SPLIT_KNOBS[4:], ids=["t16", "t100", "default", "late_big", "nofuse"])
def test_store_hot_key_in_pieces(knobs, monkeypatch):
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    n_sub = 5000
    o = orc.StoreOracle(n_sub * 18 // 4, n_sub)
    eng = _engine(W.STORE, n_rows=n_sub)
    eng.populate(n_sub)
    rng = np.random.default_rng(4)
    for n, key in ((1500, tracegen.store_key(11, 2, 8)), (5000, tracegen.store_key(11, 2, 8)), (8000, tracegen.store_key(4999, 4, 16)),
                   (6000, tracegen.store_key(77_777, 1, 0))):  # the last one is not in the table: NOT_EXIST for READ and SET
        req = tracegen.store_random(n, seed=n, n_sub_touch=n_sub, p_set=0.3, p_missing=0.05)
        hot = rng.random(n) < 0.7
        req["key"][hot] = key
        got, want = eng.submit(req), o.replay(req)
        assert got.tobytes() == want.tobytes(), n
    assert _same_rows(eng.dump_rows(0), o.dump())


@pytest.mark.parametrize("knobs", SPLIT_KNOBS[:3] + SPLIT_KNOBS[4:], ids=["t16", "t100", "default", "late_big", "nofuse"])
@pytest.mark.parametrize("same_quadrant", [False, True])
def test_tatp_hot_key_in_pieces_beside_a_neighbour_in_its_bucket(same_quadrant, knobs, monkeypatch):
    """The sub's other keys (the remainder) are resolved beside the pieces -- unless one of the hot BUCKET uses the hot key's
    lock byte (same quadrant) or restructures the chain: then the remainder says so and the whole sub goes the old way."""
    import struct

    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    n_sub = 3000
    o = orc.TatpOracle(n_sub, populate_n=n_sub, log_entries=400_000)
    hs = o.hash_size(0)
    by_bucket = {}
    for s_id in range(n_sub):
        h = orc.fasthash64(struct.pack("<Q", s_id))
        by_bucket.setdefault(h % hs, []).append((s_id, (h % (4 * hs)) // hs))
    pair = None
    for g in by_bucket.values():
        for a in g:
            for b in g:
                if a[0] < b[0] and (a[1] == b[1]) == same_quadrant:
                    pair = pair or (a[0], b[0])
    assert pair is not None
    existing = [o.dump(t)[0] for t in range(5)]
    eng = _engine(W.TATP, n_rows=n_sub, log_entries=400_000)
    eng.populate(n_sub)
    rng = np.random.default_rng(5)
    T = wire.Tatp
    for k, n in enumerate((2500, 7000, 4000)):
        req = tracegen.tatp_random(n, existing, seed=31 + k, n_sub_touch=n_sub)
        u = rng.random(n)
        for key, lo, hi in ((pair[0], 0.0, 0.70), (pair[1], 0.70, 0.76)):  # the neighbour: a few dozen requests, well under a quarter
            hot = (u >= lo) & (u < hi)
            req["table"][hot] = 0
            req["key"][hot] = key
            req["type"][hot] = rng.choice([T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.COMMIT_BCK], int(hot.sum()), p=[0.6, 0.15, 0.05, 0.1, 0.1])
        got, want = eng.submit(req), o.replay(req)
        assert got.tobytes() == want.tobytes(), k
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t)), t
    _tatp_locks(eng, o)


@pytest.mark.parametrize("no_bitmap", ["0", "1"])
@pytest.mark.parametrize("n,touch", [(60_000, 1), (90_000, 3), (30_000, 40)])
def test_smallbank_hot_accounts_bitmap_order_and_sort_agree(n, touch, no_bitmap, monkeypatch):
    """A stretch that is one account (its savings and its checking row: one or two key classes) is put in order by an
    index bitmap instead of the LDS sort, and a sub of several stretches is regrouped by stretch once
    (kv_big_bin); DINT_KV_NO_BM=1 keeps the sort.  Both against the oracle, in two passes so the second starts from
    counters the first one left."""
    monkeypatch.setenv("DINT_KV_NO_BM", no_bitmap)
    req = tracegen.sb_random(n, seed=n + touch, n_acct_touch=touch)
    eng = _engine(W.SMALLBANK, n_rows=10_000)
    eng.populate(max(touch, 8))
    o = orc.SmallbankOracle(10_000, populate_n=max(touch, 8))
    got = np.concatenate([eng.submit(req[:n // 2 + 7]), eng.submit(req[n // 2 + 7:])])
    assert got.tobytes() == o.replay(req).tobytes()
    _sb_state(eng, o)


# ---------------------------------------------------------------- the two-level partition's own corners
@pytest.mark.parametrize("knob", [("DINT_KV_CAP", "24"), ("DINT_KV_LCAP", "96"), ("DINT_KV_RPT", "1"), ("DINT_KV_RPT", "2"),
                                  ("DINT_KV_RPT", "4"), ("DINT_KV_COARSE_LOAD", "4096"), ("DINT_KV_COARSE_LOAD", "64")])
@pytest.mark.parametrize("wl", ["store", "tatp", "smallbank"])
def test_partition_corners(wl, knob, monkeypatch):
    """k_kv_part / k_kv_resolve away from their usual operating point (the knobs are read at every launch):
    DINT_KV_CAP = 24 records in place per coarse bin, so most records travel through the pass's overflow list;
    DINT_KV_LCAP = 96 records in the LDS split, so most coarse bins take the everything-is-a-big-sub fallback;
    1, 2 and 4 requests per thread of the partition kernel; coarse bins of ~4096 records (every bin spills over its LDS
    split) and of ~64 records (one chunk per bin)."""
    monkeypatch.setenv(*knob)
    if wl == "store":
        req = tracegen.store_random(70_000, seed=13, n_sub_touch=900, p_set=0.4, p_insert=0.05)
        eng = _engine(W.STORE, n_rows=5000)
        o = orc.StoreOracle(5000 * 18 // 4, 1500); eng.populate(1500)
    elif wl == "tatp":
        o = orc.TatpOracle(5000, populate_n=900)
        req = tracegen.tatp_random(70_000, [o.dump(t)[0] for t in range(5)], seed=14, n_sub_touch=900)
        eng = _engine(W.TATP, n_rows=5000); eng.populate(900)
    else:
        req = tracegen.sb_random(70_000, seed=15, n_acct_touch=900)
        eng = _engine(W.SMALLBANK, n_rows=5000); eng.populate(900)
        o = orc.SmallbankOracle(5000, populate_n=900)
    want = o.replay(req)
    got = np.concatenate([eng.submit(req[:50_000]), eng.submit(req[50_000:50_100]), eng.submit(req[50_100:])])
    assert got.tobytes() == want.tobytes()
    if wl == "tatp":
        for t in range(5):
            assert _same_rows(eng.dump_rows(t), o.dump(t)), t
        _tatp_locks(eng, o)
        ring, tail = eng.read_log(1_000_000)
        assert tail == o.tail and (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64) == o.ring).all()
    elif wl == "smallbank":
        _sb_state(eng, o)
    else:
        assert _same_rows(eng.dump_rows(0), o.dump())


# ---------------------------------------------------------------- smallbank: a hot account's row in pieces (r06, kv_sb_item)
SB_KNOBS = [{}, {"DINT_KV_SB_SPLIT_MIN": "200", "DINT_KV_SPLIT_TARGET": "64"}, {"DINT_KV_SB_SPLIT_MIN": "600", "DINT_KV_SPLIT_TARGET": "300"},
            {"DINT_KV_SB_SPLIT_MIN": "0"}, {"DINT_KV_SB_SPLIT_MIN": "2048"},
            {"DINT_KV_SB_WORKERS": "0"},   # r06a: every work item in k_kv_big (pieces included), k_kv_pass without workers
            {"DINT_KV_NO_FUSE": "1"}]      # r05's launches: k_kv_resolve -> k_kv_big


def _sb_hot(n, p_hot, mix, hot, seed, n_acct):
    """n requests; a share p_hot on the hot rows `hot` = [(table, key), ...] with op mix `mix` {type: weight}, the rest noise"""
    rng = np.random.default_rng(seed)
    req = tracegen.sb_random(n, seed=seed, n_acct_touch=n_acct)
    for tb, key in hot:  # (the noise leaves the hot rows alone)
        req["key"][(req["table"] == tb) & (req["key"] == key)] = (key + 1) % n_acct
    u = rng.random(n)
    types, w = list(mix.keys()), np.array(list(mix.values()), float)
    for k, (tb, key) in enumerate(hot):
        sel = (u >= k * p_hot / len(hot)) & (u < (k + 1) * p_hot / len(hot))
        req["table"][sel] = tb
        req["key"][sel] = key
        req["type"][sel] = rng.choice(types, int(sel.sum()), p=w / w.sum())
    return req


@pytest.mark.parametrize("knobs", SB_KNOBS, ids=["default", "t64", "t300", "off", "min2048", "big-only", "nofuse"])
@pytest.mark.parametrize("p_hot,mix,hot", [
    (0.6, {0: 30, 1: 25, 2: 15, 3: 12, 4: 10, 5: 8}, [(0, 7), (1, 7)]),          # the account's savings and checking rows, every op kind
    (0.7, {0: 40, 1: 10, 2: 38, 3: 6, 4: 6}, [(0, 3)]),                           # mostly shared traffic: long FREE stretches
    (0.7, {0: 10, 1: 45, 2: 5, 3: 38, 5: 2}, [(1, 11)]),                          # mostly exclusive: the mode changes all the time
    (0.5, {0: 20, 1: 20, 2: 25, 3: 25, 17: 10}, [(0, 5)]),                        # more releases than acquires (the counters wrap), WARMUP_READs
    (0.6, {0: 30, 1: 25, 2: 15, 3: 12, 4: 10, 5: 8}, [(0, 900_000)]),             # the hot row does not exist (missing_keys)
])
def test_smallbank_hot_row_in_pieces(p_hot, mix, hot, knobs, monkeypatch):
    """A hot account's row is answered by several workgroups at once: pieces by request index publish their op kinds as lane
    masks, one coordinator walks the shared / exclusive counters over all of them, every piece answers its requests (kv_sb_item).
    Against the oracle (smallbank/udp/server_shard.cc:121-173), pass after pass on one engine -- the counters, versions and values a
    pass leaves are what the next one starts from -- at sizes from one piece to a hundred; small thresholds force the path on
    small passes, 0 is r05's single workgroup."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    n_acct = 1_000_000 if hot[0][1] >= 1000 else 2000
    o = orc.SmallbankOracle(n_acct, populate_n=2000)
    eng = _engine(W.SMALLBANK, n_rows=n_acct)
    eng.populate(2000)
    for k, n in enumerate((3000, 900, 20_000, 6000, 70_000, 1500, 40_000)):
        req = _sb_hot(n, p_hot, mix, hot, seed=31 * k + 7, n_acct=2000)
        got, want = eng.submit(req), o.replay(req)
        assert got.tobytes() == want.tobytes(), (k, np.nonzero(np.frombuffer(got.tobytes(), "u1") != np.frombuffer(want.tobytes(), "u1"))[0][:5] // 23)
    _sb_state(eng, o)
    st = eng.stats()
    assert st["bad_requests"] == 0 and st["big_bin_requests"] > 0
    assert st["missing_keys"] == o.errors
    if hot[0][1] >= 1000:
        assert st["missing_keys"] > 1000


@pytest.mark.parametrize("share", [0.1, 0.003], ids=["warm-neighbour", "cold-neighbour"])
@pytest.mark.parametrize("knobs", SB_KNOBS[:2] + SB_KNOBS[4:], ids=["default", "t64", "min2048", "big-only", "nofuse"])
def test_smallbank_hot_row_beside_a_key_on_its_counter_pair(knobs, share, monkeypatch):
    """two accounts whose rows share a bucket AND a lock quadrant (one counter pair).  A COLD neighbour (a handful of requests per
    piece) rides in the pieces: its lock ops in the masks the coordinator walks, its row ops done by the coordinator in request
    order (kv_sb_item, r06b) -- nothing falls back.  A WARM one (more than KSB_FMAX requests in a piece) is no closed form: the
    piece says so and piece 0 takes the sub the old way (stats: late_requests)."""
    import struct

    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    n_acct = 2000
    o = orc.SmallbankOracle(n_acct, populate_n=n_acct)
    hs = o.hash_size(0)
    seen, pair = {}, None
    for a in range(n_acct):
        h = orc.fasthash64(struct.pack("<Q", a))
        slot = h % (4 * hs)
        if slot in seen and pair is None:
            pair = (seen[slot], a)
        seen.setdefault(slot, a)
    assert pair is not None
    eng = _engine(W.SMALLBANK, n_rows=n_acct)
    eng.populate(n_acct)
    rng = np.random.default_rng(3)
    for k, n in enumerate((8000, 30_000)):
        req = tracegen.sb_random(n, seed=50 + k, n_acct_touch=n_acct)
        u = rng.random(n)
        for key, lo, hi in ((pair[0], 0.0, 0.6), (pair[1], 0.6, 0.6 + share)):
            sel = (u >= lo) & (u < hi)
            req["table"][sel] = 0
            req["key"][sel] = key
            req["type"][sel] = rng.choice(6, int(sel.sum()), p=[0.3, 0.25, 0.15, 0.12, 0.1, 0.08])
        assert eng.submit(req).tobytes() == o.replay(req).tobytes(), k
    _sb_state(eng, o)
    st = eng.stats()
    assert st["missing_keys"] == o.errors
    if share >= 0.1 and knobs.get("DINT_KV_SB_SPLIT_MIN") != "200":
        assert st["late_requests"] > 0  # (pieces of ~384: ~50 of the neighbour's requests each)
