"""The CPU oracle (C restatement, run-time sizes) against the golden fixtures recorded from
the UNMODIFIED reference udp/ servers (tests/golden/*.npz, made by tests/golden/make_golden.py
at the reference's compile-time sizes).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest

import tracegen
from dint_amd import wire
from oracle import oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name, dtype):
    z = np.load(os.path.join(G, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    req = np.frombuffer(z["req"].tobytes(), dtype)
    rep = np.frombuffer(z["rep"].tobytes(), dtype)
    return z, meta, req, rep


def test_lock_fasst_golden():
    z, meta, req, rep = load("lock_fasst", wire.FASST_MSG)
    o = orc.FasstOracle(meta["nslots"])
    got = o.replay(req)
    assert got.tobytes() == rep.tobytes()
    d = np.frombuffer(z["dump"].tobytes()[4:], "<u4").reshape(-1, 3)
    nz = np.nonzero(o.locks | o.vers)[0]
    assert (d[:, 0] == nz).all() and (d[:, 1] == o.locks[nz]).all() and (d[:, 2] == o.vers[nz]).all()


def test_lock_2pl_golden():
    z, meta, req, rep = load("lock_2pl", wire.TPL_MSG)
    o = orc.TplOracle(meta["nslots"])
    got = o.replay(req)
    assert got.tobytes() == rep.tobytes()
    d = np.frombuffer(z["dump"].tobytes()[4:], "<u4").reshape(-1, 3)
    nz = np.nonzero(o.num_ex | o.num_sh)[0]
    assert (d[:, 0] == nz).all() and (d[:, 1] == o.num_ex[nz]).all() and (d[:, 2] == o.num_sh[nz]).all()


def test_log_server_golden():
    z, meta, req, rep = load("log_server", wire.LOG_MSG)
    o = orc.LogOracle(meta["ring"])
    got = o.replay(req)
    assert got.tobytes() == rep.tobytes()
    dump = z["dump"].tobytes()
    tail, n = np.frombuffer(dump, "<u4", 2)
    recs = np.frombuffer(dump, "u1", offset=8).reshape(n, 64)
    assert tail == o.tail
    assert (o.ring[:n, :52] == recs[:, :52]).all()


def test_store_golden():
    z, meta, req, rep = load("store", wire.STORE_MSG)
    o = orc.StoreOracle(meta["hash_size"], meta["touch"])
    got = o.replay(req)
    a = tracegen.mask_populate_garbage("store", got)
    b = tracegen.mask_populate_garbage("store", rep)
    assert a.tobytes() == b.tobytes()
    assert (rep["type"] == wire.Store.NOT_EXIST).sum() > 100  # the missing-key path is exercised


def test_tatp_golden():
    z, meta, req, rep = load("tatp", wire.TATP_MSG)
    o = orc.TatpOracle(meta["n_sub"], populate_n=meta["touch"])
    got = o.replay(req)
    assert o.errors == 0
    a = tracegen.mask_populate_garbage("tatp", got)
    b = tracegen.mask_populate_garbage("tatp", rep)
    assert a.tobytes() == b.tobytes()
    # final lock + log state of the reference
    tail_dump = z["dump_tail"].tobytes()
    off = 0
    for t in range(5):
        cnt = int(np.frombuffer(tail_dump, "<u4", 1, off)[0]); off += 4
        held = np.frombuffer(tail_dump, "<u4", cnt, off); off += 4 * cnt
        assert (np.nonzero(o.locks(t))[0] == held).all()
    tail, n = np.frombuffer(tail_dump, "<u4", 2, off); off += 8
    recs = np.frombuffer(tail_dump, "u1", n * 64, off).reshape(n, 64)
    assert tail == o.tail
    ring = o.ring[:n]
    is_del = recs[:, 52] == 1
    assert (ring[:, :8] == recs[:, :8]).all() and (ring[:, 48:54] == recs[:, 48:54]).all()
    assert (ring[~is_del, 8:48] == recs[~is_del, 8:48]).all()  # DELETE_LOG leaves val untouched (uninitialised there)
    seen = set(rep["type"].tolist())
    assert {4, 6, 7, 8, 9, 15, 16, 17, 20, 21, 25, 26, 27} <= seen  # every reply type occurs


def test_smallbank_golden():
    z, meta, req, rep = load("smallbank", wire.SB_MSG)
    o = orc.SmallbankOracle(meta["n_acct"], populate_n=meta["touch"])
    got = o.replay(req)
    assert o.errors == 0
    assert got.tobytes() == rep.tobytes()
    tail_dump = z["dump_tail"].tobytes()
    off = 0
    for t in range(2):
        cnt = int(np.frombuffer(tail_dump, "<u4", 1, off)[0]); off += 4
        d = np.frombuffer(tail_dump, "<u4", cnt * 3, off).reshape(cnt, 3); off += 12 * cnt
        ex, sh = o.num_ex(t), o.num_sh(t)
        nz = np.nonzero(ex | sh)[0]
        assert (d[:, 0] == nz).all() and (d[:, 1] == ex[nz]).all() and (d[:, 2] == sh[nz]).all()
    assert set(range(7, 16)) <= set(rep["type"].tolist())


@pytest.mark.skipif(not orc.ref_available("lock_fasst"), reason="reference binaries are built only where /root/reference exists")
def test_ref_server_started_ahead_of_its_trace():
    """REF_TRACE_WAIT: the reference is started first and given its trace later (bench.py's tatp reference leg)."""
    req = tracegen.fasst_random(20000, seed=5)
    want, _ = orc.ref_replay("lock_fasst", req)
    srv = orc.RefServer("lock_fasst")
    try:
        assert srv.wait_populated(60)
        got, st = srv.replay(req)
    finally:
        srv.close()
    assert got.tobytes() == want.tobytes() and st["n"] == len(req)


@pytest.mark.skipif(not orc.loopback_available(), reason="reference binaries are built only where /root/reference exists")
def test_as_shipped_reference_server_over_loopback():
    """BASELINE.md 3(2) plumbing: the unmodified lock_fasst/udp/server (real UDP sockets, bind redirected to 127.0.0.1)
    answers the closed-loop client; a short run, only the mechanics are checked here."""
    req = tracegen.fasst_random(50_000, seed=9)
    try:
        r = orc.ref_loopback_fasst(req, server_threads=2, client_threads=2, window=8, warmup_s=0.2, measure_s=0.6)
    except RuntimeError as ex:  # the reference's port is hard-coded (20230): busy on this host
        pytest.skip(str(ex))
    assert r["replies"] > 1000 and r["server_threads"] == 2 and r["ops_per_s"] > 0
