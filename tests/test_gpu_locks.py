"""GPU parity: lock_fasst / lock_2pl / log_server through the C ABI vs the CPU oracle and
the golden fixtures recorded from the unmodified reference.  Bit-exact (integer work)."""
import json
import os

import numpy as np
import pytest
import torch

import tracegen
from dint_amd import wire
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _engine(*a, **k):
    from dint_amd.engine import Engine

    return Engine(*a, **k)


def _golden(name, dtype):
    z = np.load(os.path.join(G, name + ".npz"))
    return z, json.loads(str(z["meta"])), np.frombuffer(z["req"].tobytes(), dtype), np.frombuffer(z["rep"].tobytes(), dtype)


def test_fasst_golden_reference_sizes():
    z, meta, req, rep = _golden("lock_fasst", wire.FASST_MSG)
    eng = _engine(wire.Workload.FASST, n_slots=meta["nslots"])
    got = eng.submit(req)
    assert got.tobytes() == rep.tobytes()
    lock, ver = eng.read_locks()
    d = np.frombuffer(z["dump"].tobytes()[4:], "<u4").reshape(-1, 3)
    nz = np.nonzero(lock | ver)[0]
    assert (d[:, 0] == nz).all() and (d[:, 1] == lock[nz]).all() and (d[:, 2] == ver[nz]).all()


def test_fasst_kat2():
    F = wire.Fasst
    ops = [F.ACQUIRE_LOCK, F.ACQUIRE_LOCK, F.READ, F.COMMIT, F.READ, F.ACQUIRE_LOCK, F.ABORT, F.ACQUIRE_LOCK]
    m = np.zeros(len(ops), wire.FASST_MSG)
    m["type"], m["lid"] = ops, 7
    r = _engine(wire.Workload.FASST).submit(m)
    assert r["type"].tolist() == [5, 6, 4, 8, 4, 5, 7, 5] and r["ver"][2] == 0 and r["ver"][4] == 1


@pytest.mark.parametrize("n,nslots,n_hot,p_hot", [
    (1, 1 << 20, 4, 0.5), (63, 1 << 20, 4, 0.9), (64, 97, 4, 0.5), (65, 1 << 20, 1, 1.0),
    (4096, 1 << 20, 16, 0.8), (65536, 1 << 20, 64, 0.7), (65536, 36_000_000, 3, 0.95),
    (65536, 1 << 20, 1, 1.0), (200_000, 4800, 64, 0.3), (65537, 1, 2, 0.5),
])
def test_fasst_vs_oracle(n, nslots, n_hot, p_hot):
    req = tracegen.fasst_random(n, seed=n + nslots, n_hot=n_hot, p_hot=p_hot)
    eng = _engine(wire.Workload.FASST, n_slots=nslots)
    o = orc.FasstOracle(nslots)
    got = eng.submit(req)
    want = o.replay(req)
    assert got.tobytes() == want.tobytes()
    lock, ver = eng.read_locks()
    assert (lock == o.locks).all() and (ver == o.vers).all()


def test_fasst_batch_split_invariance():
    """The reply stream must not depend on how the trace is cut into batches."""
    req = tracegen.fasst_random(50_000, seed=3, n_hot=32, p_hot=0.8)
    want = orc.FasstOracle(1 << 20).replay(req)
    for bs in (1, 64, 4096, 65536):
        if bs == 1:
            sub = req[:300]
            eng = _engine(wire.Workload.FASST, n_slots=1 << 20)
            got = np.concatenate([eng.submit(sub[i:i + 1]) for i in range(len(sub))])
            assert got.tobytes() == want[:300].tobytes()
            continue
        eng = _engine(wire.Workload.FASST, n_slots=1 << 20)
        got = np.concatenate([eng.submit(req[i:i + bs]) for i in range(0, len(req), bs)])
        assert got.tobytes() == want.tobytes(), bs


def test_fasst_bad_types_are_echoed_and_counted():
    req = tracegen.fasst_random(5000, seed=9)
    req["type"][::7] = 200
    eng = _engine(wire.Workload.FASST, n_slots=1 << 16)
    o = orc.FasstOracle(1 << 16)
    got, want = eng.submit(req), o.replay(req)
    assert got.tobytes() == want.tobytes()
    assert eng.stats()["bad_requests"] == o.errors == len(req[::7])


def test_fasst_device_inplace_and_empty():
    import torch

    eng = _engine(wire.Workload.FASST, n_slots=1 << 20)
    assert len(eng.submit(np.zeros(0, wire.FASST_MSG))) == 0
    req = tracegen.fasst_random(70_000, seed=21, n_hot=8, p_hot=0.6)
    d = torch.from_numpy(np.frombuffer(req.tobytes(), np.uint8).copy()).cuda()
    eng.submit_device(d, len(req), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = np.frombuffer(d.cpu().numpy().tobytes(), wire.FASST_MSG)
    assert got.tobytes() == orc.FasstOracle(1 << 20).replay(req).tobytes()


def test_tpl_golden_reference_sizes():
    z, meta, req, rep = _golden("lock_2pl", wire.TPL_MSG)
    eng = _engine(wire.Workload.TPL, n_slots=meta["nslots"])
    assert eng.submit(req).tobytes() == rep.tobytes()
    ex, sh = eng.read_locks()
    d = np.frombuffer(z["dump"].tobytes()[4:], "<u4").reshape(-1, 3)
    nz = np.nonzero(ex | sh)[0]
    assert (d[:, 0] == nz).all() and (d[:, 1] == ex[nz]).all() and (d[:, 2] == sh[nz]).all()


@pytest.mark.parametrize("n,nslots,n_hot,p_hot", [
    (1, 1 << 20, 4, 0.5), (100, 7, 4, 0.5), (4096, 1 << 20, 16, 0.8), (65536, 36_000_000, 64, 0.7),
    (65536, 1 << 20, 1, 1.0), (150_000, 1 << 20, 200, 0.9),
])
def test_tpl_vs_oracle(n, nslots, n_hot, p_hot):
    req = tracegen.tpl_random(n, seed=n + nslots, n_hot=n_hot, p_hot=p_hot)
    req["type"][5::11] = 9  # unknown lock types: acquire -> bad request, release -> ack only
    eng = _engine(wire.Workload.TPL, n_slots=nslots)
    o = orc.TplOracle(nslots)
    got, want = eng.submit(req), o.replay(req)
    assert got.tobytes() == want.tobytes()
    ex, sh = eng.read_locks()
    assert (ex == o.num_ex).all() and (sh == o.num_sh).all()
    assert eng.stats()["bad_requests"] == o.errors


def test_log_golden():
    z, meta, req, rep = _golden("log_server", wire.LOG_MSG)
    eng = _engine(wire.Workload.LOG, log_entries=meta["ring"])
    assert eng.submit(req).tobytes() == rep.tobytes()
    dump = z["dump"].tobytes()
    tail, n = np.frombuffer(dump, "<u4", 2)
    recs = np.frombuffer(dump, "u1", offset=8).reshape(n, 64)
    ring, t = eng.read_log(int(n))
    assert t == tail
    assert (np.frombuffer(ring.tobytes(), "u1").reshape(n, 64)[:, :52] == recs[:, :52]).all()


@pytest.mark.parametrize("n,cap", [(1, 1000), (999, 1000), (1000, 1000), (70_000, 1000), (70_000, 1_000_000), (200_001, 65_537)])
def test_log_vs_oracle_with_wrap(n, cap):
    req = tracegen.log_random(n, seed=n)
    req["type"][3::17] = 5  # not a COMMIT: the reference panics; we echo and count
    eng = _engine(wire.Workload.LOG, log_entries=cap)
    o = orc.LogOracle(cap)
    for lo in range(0, n, 50_000):  # several submits: the tail carries over
        got = eng.submit(req[lo:lo + 50_000])
        want = o.replay(req[lo:lo + 50_000])
        assert got.tobytes() == want.tobytes()
    ring, t = eng.read_log(cap)
    assert t == o.tail
    assert (np.frombuffer(ring.tobytes(), "u1").reshape(cap, 64) == o.ring).all()


# ---------------------------------------------------------------- passes of up to 2^20 requests, hot slots
@pytest.mark.parametrize("wl", ["fasst", "tpl"])
@pytest.mark.parametrize("n,nslots,n_hot,p_hot", [
    (1 << 20, 36_000_000, 3, 0.02),     # one pass; three slots with ~7,000 requests each: several stretches, chunk-crossing walks
    (1 << 20, 97, 64, 0.5),             # 97 slots: every bin is big, slots of ~5,000 and ~20,000 requests
    (300_000, 1 << 20, 2000, 0.9),      # many warm slots (~135 requests each): big bins with dozens of slots per chunk
    (1_200_000, 1 << 20, 1, 0.5),       # two passes; 600,000 requests on ONE slot
])
def test_locks_big_passes_vs_oracle(wl, n, nslots, n_hot, p_hot):
    if wl == "fasst":
        req, W_, mk = tracegen.fasst_random(n, seed=n_hot, n_hot=n_hot, p_hot=p_hot), wire.Workload.FASST, orc.FasstOracle
    else:
        req, W_, mk = tracegen.tpl_random(n, seed=n_hot, n_hot=n_hot, p_hot=p_hot), wire.Workload.TPL, orc.TplOracle
    eng, o = _engine(W_, n_slots=nslots), mk(nslots)
    got, want = eng.submit(req), o.replay(req)
    assert got.tobytes() == want.tobytes()
    a, b = eng.read_locks()
    if wl == "fasst":
        assert (a == o.locks).all() and (b == o.vers).all()
    else:
        assert (a == o.num_ex).all() and (b == o.num_sh).all()
    # the same trace cut into small passes gives the same replies
    eng2 = _engine(W_, n_slots=nslots, max_pass=4096)
    assert eng2.submit(req[:200_000]).tobytes() == want[:200_000].tobytes()


@pytest.mark.parametrize("n,max_pass,p_hot,p_acq", [
    (65536, 0, 0.9, 0.5),      # one hot slot holds 90 % of a 64k pass: the dominant-slot path (index bitmap + mode walk)
    (65536, 0, 0.6, 0.8),      # ... mostly ACQUIREs (the lock is held most of the time)
    (65536, 0, 1.0, 0.5),      # nothing but the hot slot
    (200_000, 65536, 0.85, 0.55),   # several passes, the counters carry over
    (200_000, 0, 0.85, 0.55),  # one pass of more than 65,536 requests: the general big-bin path (sorted stretches + walk)
    (20_000, 0, 0.3, 0.5),     # a dominant slot of a few thousand among other traffic
    (65536, 0, 0.06, 0.5),     # ... of 4,000 in a 64k pass: three requests per 64-bit word of the index bitmaps
    (65536, 0, 0.2, 0.93),     # the bench shape: a fifth of the pass retries one lock, nearly all of it ACQUIREs
    (60_001, 0, 0.5, 0.5),     # a pass that ends inside a bitmap word
])
def test_2pl_dominant_slot_vs_oracle(n, max_pass, p_hot, p_acq):
    """the closed-loop shape that makes a lock hot (hundreds of workers retrying it), plus releases nobody holds (the
    reference's unsigned counters wrap: lock_2pl/udp/server.cc:108-111)"""
    rng = np.random.default_rng(n + int(100 * p_hot))
    m = np.zeros(n, wire.TPL_MSG)
    m["action"] = (rng.random(n) >= p_acq).astype(np.uint8)
    m["type"] = (rng.random(n) < 0.3).astype(np.uint8)
    hot = rng.random(n) < p_hot
    m["lid"] = np.where(hot, 777, rng.integers(0, 24_000_000, n)).astype("<u4")
    m["lid"][rng.random(n) < 0.02] = 778  # a second, smaller hot slot
    m["type"][(rng.random(n) < 0.01) & (m["action"] == 1)] = 7  # RELEASEs of an unknown lock type: acked, nothing changes (server.cc:104-118)
    eng = _engine(wire.Workload.TPL, n_slots=1 << 20, max_pass=max_pass)
    o = orc.TplOracle(1 << 20)
    assert eng.submit(m).tobytes() == o.replay(m).tobytes()
    ex, sh = eng.read_locks()
    assert (ex == o.num_ex).all() and (sh == o.num_sh).all()


@pytest.mark.parametrize("wl", ["fasst", "tpl"])
def test_dominant_slot_when_the_samples_name_another_slot(wl, monkeypatch):
    """the dominant-slot path fills its index bitmaps for the majority of eight sampled records while it counts the
    candidates; DINT_LOCK_MISGUESS makes that guess wrong every time, so the bitmaps are cleared and filled again"""
    monkeypatch.setenv("DINT_LOCK_MISGUESS", "1")
    n = 65536
    if wl == "fasst":
        req, W_, mk = tracegen.fasst_random(n, seed=5, n_hot=2, p_hot=0.5), wire.Workload.FASST, orc.FasstOracle
    else:
        req, W_, mk = tracegen.tpl_random(n, seed=5, n_hot=2, p_hot=0.5), wire.Workload.TPL, orc.TplOracle
    eng, o = _engine(W_, n_slots=1 << 20), mk(1 << 20)
    assert eng.submit(req).tobytes() == o.replay(req).tobytes()
    a, b = eng.read_locks()
    if wl == "fasst":
        assert (a == o.locks).all() and (b == o.vers).all()
    else:
        assert (a == o.num_ex).all() and (b == o.num_sh).all()


@pytest.mark.parametrize("wl", ["fasst", "tpl"])
def test_inputs_ready_passes_overlap_and_match_the_oracle(wl):
    """DINT_FLAG_INPUTS_READY: the first half of a lock pass (k_lock_count, k_kv_scan_place) runs on the engine's helper stream
    beside the previous pass's resolve kernel, three scratch sets in turn.  Forty batches back to back -- separate reply
    buffers, replies in place, a call of several passes, caller streams, a stretch with kernel timing on (one stream), a
    snapshot / restore in between -- and every reply byte and the final table equal the oracle's."""
    from dint_amd import _lib
    n, nb = 40_000, 40
    if wl == "fasst":
        mk_req, W_, mk, msg = tracegen.fasst_random, wire.Workload.FASST, orc.FasstOracle, wire.FASST_MSG.itemsize
    else:
        mk_req, W_, mk, msg = tracegen.tpl_random, wire.Workload.TPL, orc.TplOracle, wire.TPL_MSG.itemsize
    reqs = [mk_req(n, seed=100 + k, n_hot=3, p_hot=0.4) for k in range(nb)]
    eng, o = _engine(W_, n_slots=1 << 16, flags=_lib.FLAG_INPUTS_READY, max_pass=16_384), mk(1 << 16)  # three passes per call
    want = [o.replay(r) for r in reqs]
    d_req = [torch.from_numpy(np.frombuffer(r.tobytes(), np.uint8).copy()).cuda() for r in reqs]
    d_rep = [torch.empty_like(x) for x in d_req]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    eng.snapshot()
    for rnd in range(2):  # the second round after a restore: the same replies again
        for k in range(nb):
            if k == 17:
                eng.timing_enable(True)  # passes on one stream in between
            if k == 21:
                eng.timing_enable(False)
            out = d_req[k] if (k % 5 == 4 and rnd == 1) else d_rep[k]  # in place (the requests are not needed again)
            st = 0 if k % 3 == 0 else streams[k & 1].cuda_stream
            eng.submit_device(d_req[k], n, out, st)
        torch.cuda.synchronize()
        eng.sync()
        for k in range(nb):
            got = (d_req[k] if (k % 5 == 4 and rnd == 1) else d_rep[k]).cpu().numpy().tobytes()
            assert got == want[k].tobytes(), (rnd, k)
        a, b = eng.read_locks()
        if wl == "fasst":
            assert (a == o.locks).all() and (b == o.vers).all()
        else:
            assert (a == o.num_ex).all() and (b == o.num_sh).all()
        if rnd == 0:
            eng.restore()


@pytest.mark.parametrize("wl", ["fasst", "tpl"])
def test_direct_big_bins_and_the_overflow_list_agree(wl, monkeypatch):
    """a pass of <= 65,536 requests stores a big bin's records beyond the 64 in place straight into a region of its own (two
    launches); larger passes, and DINT_LOCK_NO_DIRECT, take the overflow list and k_kv_scan_place (three).  One engine goes
    through both kinds of pass in turn -- the region names must not survive a pass -- and an engine without regions replays
    the same stream; every reply byte against the oracle."""
    if wl == "fasst":
        mk_req, W_, mk = tracegen.fasst_random, wire.Workload.FASST, orc.FasstOracle
    else:
        mk_req, W_, mk = tracegen.tpl_random, wire.Workload.TPL, orc.TplOracle
    sizes = [65536, 200_000, 30_000, 65536, 100_000, 500, 65536]
    reqs = [mk_req(n, seed=7 + k, n_hot=5, p_hot=0.5) for k, n in enumerate(sizes)]
    o = mk(1 << 18)
    want = [o.replay(r) for r in reqs]
    eng = _engine(W_, n_slots=1 << 18)  # passes of up to 2^20: 200,000 requests are ONE pass (the overflow list), 65,536 a direct one
    for r, w in zip(reqs, want):
        assert eng.submit(r).tobytes() == w.tobytes()
    monkeypatch.setenv("DINT_LOCK_NO_DIRECT", "1")
    eng2 = _engine(W_, n_slots=1 << 18)
    for r, w in zip(reqs, want):
        assert eng2.submit(r).tobytes() == w.tobytes()
    for e in (eng, eng2):
        a, b = e.read_locks()
        if wl == "fasst":
            assert (a == o.locks).all() and (b == o.vers).all()
        else:
            assert (a == o.num_ex).all() and (b == o.num_sh).all()


def test_2pl_client_trace_vs_oracle():
    """the lock_2pl client loop (dint_amd.driver.TplClient) against the engine, 64k-request passes; the oracle replays it"""
    from dint_amd.driver import tpl_trace

    eng = _engine(wire.Workload.TPL, n_slots=1 << 20, max_pass=65536)
    req, rep, st = tpl_trace(eng, 65536 * 12, n_workers=4096, zipf_theta=0.8)
    assert st["protocol_errors"] == 0 and st["committed"] > 1000 and st["rejects"] > 1000
    o = orc.TplOracle(1 << 20)
    assert rep.tobytes() == o.replay(req).tobytes()
    # the replay in 64k passes from the empty table meets the dominant-slot path with a slot of thousands of requests
    eng2 = _engine(wire.Workload.TPL, n_slots=1 << 20, max_pass=65536)
    assert eng2.submit(req).tobytes() == rep.tobytes()
    ex, sh = eng2.read_locks()
    assert (ex == o.num_ex).all() and (sh == o.num_sh).all()


def test_device_primitives_match_their_portable_forms():
    """dint_selftest: the wave sort network built from DPP / v_permlane*_swap exchanges (dint_device.h) against the
    ds_bpermute network, and every single lane exchange against __shfl_xor."""
    from dint_amd import _lib

    assert _lib.load().dint_selftest(-1) == 0
