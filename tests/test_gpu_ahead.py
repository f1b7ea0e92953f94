"""GPU parity of the look-ahead passes (round 6: dint_submit_device_ahead, k_kv_hot_part): the partition stage of batch
k + 1 runs in the same launch as the hot keys of batch k.  Replies, rows, lock words and log ring must be exactly those
of the serial reference (the oracle, pinned to tatp/udp/server_shard.cc:113-210 and store/udp/server.cc:75-97) -- over
dozens of back-to-back passes with INSERT / DELETE (chain entries allocated, freed and recycled across passes: the two
pend sets of kv_pool_rotate), on three engines at once, in place and into separate reply buffers."""
import numpy as np
import pytest

import tracegen
from dint_amd import _lib, wire
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
W = wire.Workload


def _engine(*a, **k):
    from dint_amd.engine import Engine

    return Engine(*a, **k)


def _same_rows(a, b):
    return all(x.shape == y.shape and (x == y).all() for x, y in zip(a, b))


def _up(a):
    import torch

    return torch.from_numpy(np.frombuffer(a.tobytes(), np.uint8).copy()).cuda()


def _cuts(n, k, rng, lo=1):
    """k batch sizes adding up to n, uneven (some tiny, some large)"""
    w = rng.random(k) ** 2 + 0.01
    c = np.maximum(lo, (w / w.sum() * n).astype(np.int64))
    c[-1] = max(lo, n - int(c[:-1].sum()))
    assert c.sum() >= n
    off = np.concatenate([[0], np.cumsum(c)])
    return [(int(off[i]), int(min(n, off[i + 1]))) for i in range(k) if off[i] < n]


def _chain(eng, d_req, sizes, d_rep=None, stream=0):
    """submit_device over consecutive batches, each call announcing the next"""
    for i, (d, n) in enumerate(zip(d_req, sizes)):
        nxt = None
        if i + 1 < len(sizes):
            nxt = (d_req[i + 1], sizes[i + 1], None if d_rep is None else d_rep[i + 1])
        eng.submit_device(d, n, None if d_rep is None else d_rep[i], stream, ahead=nxt)


def _tatp_check_state(eng, o, cap):
    for t in range(5):
        assert _same_rows(eng.dump_rows(t), o.dump(t)), t
        lk, _ = eng.read_locks(t)
        assert (lk == o.locks(t)).all(), t
    ring, tail = eng.read_log(cap)
    assert tail == o.tail
    assert (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64) == o.ring).all()


@pytest.mark.parametrize("inplace", [True, False])
@pytest.mark.parametrize("n,k,n_sub,touch", [(240_000, 48, 4000, 60), (600_000, 41, 50_000, 3000), (90_000, 44, 2000, 3)])
def test_tatp_many_passes_with_look_ahead(n, k, n_sub, touch, inplace):
    import torch

    rng = np.random.default_rng(n + k)
    o = orc.TatpOracle(n_sub, log_entries=100_000, populate_n=touch)
    existing = [o.dump(t)[0] for t in range(5)]
    req = tracegen.tatp_random(n, existing, seed=n + touch, n_sub_touch=touch)  # READ / ACQUIRE / ABORT / COMMIT / INSERT / DELETE / logs
    eng = _engine(W.TATP, n_rows=n_sub, log_entries=100_000)
    eng.populate(touch)
    cuts = _cuts(n, k, rng)
    d_req = [_up(req[a:b]) for a, b in cuts]
    d_rep = None if inplace else [torch.empty_like(d) for d in d_req]
    _chain(eng, d_req, [b - a for a, b in cuts], d_rep)
    eng.sync()
    got = np.concatenate([(d_req if inplace else d_rep)[i].cpu().numpy() for i in range(len(cuts))])
    want = o.replay(req)
    assert got.tobytes() == want.tobytes()
    if not inplace:
        assert np.concatenate([d.cpu().numpy() for d in d_req]).tobytes() == req.tobytes()  # requests untouched
    _tatp_check_state(eng, o, 100_000)
    st = eng.stats()
    assert st["bad_requests"] + st["missing_keys"] == o.errors and st["pool_exhausted"] == 0
    assert st["batches"] == len(cuts)


def test_store_many_passes_with_look_ahead_and_one_long_submit():
    """store with INSERTs, 40 announced batches; then the same trace as ONE submit_device of several passes (max_pass
    4096: the passes of one call look ahead at each other without any announcement)"""
    n, n_sub = 200_000, 3000
    req = tracegen.store_random(n, seed=3, n_sub_touch=80, p_set=0.4, p_insert=0.08)
    want_o = orc.StoreOracle(n_sub * 18 // 4, 160)
    want = want_o.replay(req)
    rng = np.random.default_rng(5)
    cuts = _cuts(n, 40, rng)
    eng = _engine(W.STORE, n_rows=n_sub)
    eng.populate(160)
    d_req = [_up(req[a:b]) for a, b in cuts]
    _chain(eng, d_req, [b - a for a, b in cuts])
    eng.sync()
    assert np.concatenate([d.cpu().numpy() for d in d_req]).tobytes() == want.tobytes()
    assert _same_rows(eng.dump_rows(0), want_o.dump())
    eng2 = _engine(W.STORE, n_rows=n_sub, max_pass=4096)
    eng2.populate(160)
    d = _up(req)
    eng2.submit_device(d, n)
    eng2.sync()
    assert d.cpu().numpy().tobytes() == want.tobytes()
    assert _same_rows(eng2.dump_rows(0), want_o.dump())
    assert eng2.stats()["batches"] == (n + 4095) // 4096


def test_three_engines_alternating_with_look_ahead():
    """three tatp shard servers on their own streams (bench.py's replay): each engine's chain is resolve -> hot + the
    next batch's partition; every engine against its own oracle"""
    n, k, n_sub, touch = 150_000, 40, 3000, 50
    engs, oracles, reqs, bufs, cutss = [], [], [], [], []
    for s in range(3):
        o = orc.TatpOracle(n_sub, log_entries=100_000, populate_n=touch)
        existing = [o.dump(t)[0] for t in range(5)]
        req = tracegen.tatp_random(n, existing, seed=100 + s, n_sub_touch=touch)
        e = _engine(W.TATP, n_rows=n_sub, log_entries=100_000)
        e.populate(touch)
        cuts = _cuts(n, k, np.random.default_rng(s))
        engs.append(e); oracles.append(o); reqs.append(req); cutss.append(cuts)
        bufs.append([_up(req[a:b]) for a, b in cuts])
    for i in range(max(len(c) for c in cutss)):  # epoch by epoch, the three servers side by side
        for s in range(3):
            if i < len(cutss[s]):
                a, b = cutss[s][i]
                nxt = None
                if i + 1 < len(cutss[s]):
                    a2, b2 = cutss[s][i + 1]
                    nxt = (bufs[s][i + 1], b2 - a2, None)
                engs[s].submit_device(bufs[s][i], b - a, None, 0, ahead=nxt)
    for s in range(3):
        engs[s].sync()
        got = np.concatenate([d.cpu().numpy() for d in bufs[s]])
        assert got.tobytes() == oracles[s].replay(reqs[s]).tobytes(), s
        _tatp_check_state(engs[s], oracles[s], 100_000)


def test_tatp_insert_delete_recycling_with_a_small_pool():
    """INSERT_PRIM / DELETE_PRIM of ~360 CALL_FORWARDING rows into a table of 56 buckets, pass after pass: every insert
    pass allocates ~60 chain entries, every delete pass frees them.  120 passes would need ~3,600 entries without
    recycling; the pool has 1,500 -- the run only works if freed entries come back across the look-ahead passes (two
    pend sets, kv_pool_rotate: frees of pass k are poppable from pass k + 2 on)."""
    T = wire.Tatp
    n_sub, touch, rounds = 40, 30, 120
    o = orc.TatpOracle(n_sub, log_entries=50_000, populate_n=touch)
    # CALL_FORWARDING (table 4) keys of subscribers that exist: s_id | sf_type << 32 | start_time << 40
    keys = np.array([s | (sf << 32) | (st << 40) for s in range(touch) for sf in (1, 2, 3, 4) for st in (0, 8, 16)], np.uint64)
    batches = []
    for r in range(rounds):
        m = np.zeros(len(keys), wire.TATP_MSG)
        m["table"] = 4
        m["key"] = keys
        m["type"] = T.INSERT_PRIM if r % 2 == 0 else T.DELETE_PRIM
        m["val"] = (r * 7) & 0xFF
        batches.append(m)
    want = o.replay(np.concatenate(batches))
    eng = _engine(W.TATP, n_rows=n_sub, log_entries=50_000, pool_entries=1500)
    eng.populate(touch)
    d = [_up(x) for x in batches]
    _chain(eng, d, [len(x) for x in batches])
    eng.sync()
    st = eng.stats()
    got = np.concatenate([x.cpu().numpy() for x in d])
    assert st["pool_exhausted"] == 0
    assert got.tobytes() == want.tobytes()
    _tatp_check_state(eng, o, 50_000)


def test_a_broken_announcement_is_an_error_and_the_engine_stays_usable():
    """the announced batch must be the next submission: anything else is DINT_ESTATE; the scratch is clean afterwards
    (store: no log, so the engine's state is exactly that of the batches it answered)"""
    req = tracegen.store_random(60_000, seed=8, n_sub_touch=40, p_set=0.4, p_insert=0.05)
    o = orc.StoreOracle(1000 * 18 // 4, 80)
    eng = _engine(W.STORE, n_rows=1000)
    eng.populate(80)
    a, b, c = _up(req[:20_000]), _up(req[20_000:40_000]), _up(req[40_000:])
    eng.submit_device(a, 20_000, None, 0, ahead=(b, 20_000, None))
    with pytest.raises(_lib.DintError, match="announced"):
        eng.submit_device(c, 20_000)
    eng.submit_device(b, 20_000)         # no announcement pending any more: a plain pass (the partition runs again)
    eng.submit_device(c, 20_000, None, 0, ahead=(a, 0, None))  # an empty announcement is none
    eng.sync()
    got = np.concatenate([x.cpu().numpy() for x in (a, b, c)])
    assert got.tobytes() == o.replay(req).tobytes()
    assert _same_rows(eng.dump_rows(0), o.dump())
    # reset / restore drop an announcement silently
    eng.snapshot()
    d = _up(req[:20_000])
    eng.submit_device(d, 20_000, None, 0, ahead=(b, 20_000, None))
    eng.restore()
    d2 = _up(req[:1000])
    eng.submit_device(d2, 1000)
    eng.sync()
    o2 = orc.StoreOracle(1000 * 18 // 4, 80)
    o2.replay(req)
    assert d2.cpu().numpy().tobytes() == o2.replay(req[:1000]).tobytes()


def test_smallbank_and_lock_tables_ignore_the_announcement():
    req = tracegen.sb_random(80_000, seed=4, n_acct_touch=30)
    o = orc.SmallbankOracle(10_000, populate_n=60)
    eng = _engine(W.SMALLBANK, n_rows=10_000)
    eng.populate(60)
    cuts = _cuts(len(req), 12, np.random.default_rng(2))
    d = [_up(req[a:b]) for a, b in cuts]
    _chain(eng, d, [b - a for a, b in cuts])
    eng.sync()
    assert np.concatenate([x.cpu().numpy() for x in d]).tobytes() == o.replay(req).tobytes()
    reqf = tracegen.fasst_random(50_000, seed=5, n_hot=16, p_hot=0.8)
    ef = _engine(W.FASST, n_slots=1 << 20)
    df = [_up(reqf[:30_000]), _up(reqf[30_000:])]
    _chain(ef, df, [30_000, 20_000])
    ef.sync()
    assert np.concatenate([x.cpu().numpy() for x in df]).tobytes() == orc.FasstOracle(1 << 20).replay(reqf).tobytes()


@pytest.mark.parametrize("wl", ["fasst", "tpl"])
@pytest.mark.parametrize("inplace", [True, False])
def test_lock_tables_many_passes_with_look_ahead(wl, inplace):
    """lock_fasst / lock_2pl: the count stage of batch k + 1 rides in the resolve launch of batch k (k_lock_pass; two scratch
    sets in turn).  60 announced batches of uneven size with hot slots (direct big bins, the dominant-slot path) against the
    oracle (lock_fasst/udp/server.cc:78-119, lock_2pl/udp/server.cc:70-122); then ONE submit_device of 300,000 requests -- five
    passes of 65,536 that look ahead at each other; then a broken announcement."""
    import torch

    n = 400_000
    if wl == "fasst":
        req, o = tracegen.fasst_random(n, seed=21, n_hot=6, p_hot=0.5), orc.FasstOracle(1 << 20)
        eng = _engine(W.FASST, n_slots=1 << 20)
    else:
        req, o = tracegen.tpl_random(n, seed=22, n_hot=6, p_hot=0.5), orc.TplOracle(1 << 20)
        eng = _engine(W.TPL, n_slots=1 << 20)
    assert eng.pass_max == 65536
    want = o.replay(req)
    cuts = [c for c in _cuts(100_000, 60, np.random.default_rng(4))]
    d_req = [_up(req[a:b]) for a, b in cuts]
    d_rep = None if inplace else [torch.empty_like(d) for d in d_req]
    _chain(eng, d_req, [b - a for a, b in cuts], d_rep)
    big = _up(req[100_000:])
    eng.submit_device(big, 300_000)  # (passes of 65,536: the slices of one call announce each other)
    eng.sync()
    got = np.concatenate([(d_req if inplace else d_rep)[i].cpu().numpy() for i in range(len(cuts))] + [big.cpu().numpy()])
    assert got.tobytes() == want.tobytes()
    a, b = eng.read_locks()
    if wl == "fasst":
        assert (a == o.locks).all() and (b == o.vers).all()
    else:
        assert (a == o.num_ex).all() and (b == o.num_sh).all()
    assert eng.stats()["batches"] == len(cuts) + 5
    # a broken announcement: DINT_ESTATE, and the engine goes on from a clean scratch
    more = tracegen.fasst_random(30_000, seed=23, n_hot=4, p_hot=0.5) if wl == "fasst" else tracegen.tpl_random(30_000, seed=23, n_hot=4, p_hot=0.5)
    x, y, z = _up(more[:10_000]), _up(more[10_000:20_000]), _up(more[20_000:])
    eng.submit_device(x, 10_000, None, 0, ahead=(y, 10_000, None))
    with pytest.raises(_lib.DintError, match="announced"):
        eng.submit_device(z, 10_000)
    y = _up(more[10_000:20_000])  # (the announced batch's buffers are undefined after the error: its count stage wrote default reply codes)
    eng.submit_device(y, 10_000)
    eng.submit_device(z, 10_000)
    eng.sync()
    assert np.concatenate([t.cpu().numpy() for t in (x, y, z)]).tobytes() == o.replay(more).tobytes()
