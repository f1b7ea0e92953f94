"""Long reference-made traces (tests/golden/long_traces.json, made by tests/golden/make_long.py): 3,000,000 requests each
for lock_2pl, log_server, store, smallbank and tatp, replayed through the UNMODIFIED reference udp/ servers at their
compile-time sizes; hashes of the reply streams are committed.  The traces are regenerated here (tests/long_traces.py)
-- the closed-loop ones through the servers under test, so a single wrong grant would send the clients down another
path and change the request hash -- and compared: the CPU oracle (not gpu), the engines at two pass sizes (gpu)."""
import hashlib
import json
import os

import numpy as np
import pytest

import long_traces as lt
from dint_amd import wire
from oracle import oracle as orc

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "long_traces.json")))
W = wire.Workload


def sha(a) -> str:
    return hashlib.sha256(a.tobytes()).hexdigest()


def _check(wl, req, rep):
    f = FIX[wl]
    assert len(req) == f["n_requests"] and sha(req) == f["req_sha256"], "the trace itself differs (a reply sent a client down another path, or a generator drifted)"
    rep = orc.mask_populate_garbage(wl, rep) if wl in ("store", "smallbank", "tatp") else rep
    assert sha(rep[:1 << 20]) == f["rep_prefix_1m_sha256"]
    assert sha(rep) == f["rep_sha256"]
    assert lt.reply_types(wl, rep) == f["reply_types"]


def _check_state(wl, digest):
    """the state the 3M requests leave behind, in canonical form (long_traces.digest_of_*: rows in bucket / chain order, lock
    words, log ring) -- the unmodified reference's dump hashed the same way when the fixture was made"""
    assert digest == FIX[wl]["dump_sha256"], f"{wl}: the final state differs from the unmodified reference's"


@pytest.mark.parametrize("wl", ["lock_2pl", "log_server", "store"])
def test_oracle_long_trace(wl):
    servers = lt.oracle_servers(wl)
    req, rep = lt.TRACES[wl](servers)
    _check(wl, req, rep)
    if wl == "store":
        _check_state(wl, lt.digest_of_oracle(wl, servers[0].o))


@pytest.mark.slow
def test_oracle_long_trace_smallbank():
    req, rep = lt.TRACES["smallbank"](lt.oracle_servers("smallbank"))
    _check("smallbank", req, rep)
    srv = lt.fresh_oracle("smallbank")  # (the closed loop ran its last epoch past request N)
    srv.submit(req)
    _check_state("smallbank", lt.digest_of_oracle("smallbank", srv.o))


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("DINT_LONG_TATP") != "1", reason="three 7M-subscriber CPU oracles: ~10 minutes of page faults "
                    "in the build container (set DINT_LONG_TATP=1); tests/golden/make_long.py ran it when the fixture was made")
def test_oracle_long_trace_tatp():
    req, rep = lt.TRACES["tatp"](lt.oracle_servers("tatp"))
    _check("tatp", req, rep)
    srv = lt.fresh_oracle("tatp")
    srv.submit(req)
    _check_state("tatp", lt.digest_of_oracle("tatp", srv.o, lt.n_log_appends("tatp", rep)))


# ------------------------------------------------------------------------------------------------- GPU
def _replay(eng, req, batch):
    return np.concatenate([eng.submit(req[i:i + batch]) for i in range(0, len(req), batch)])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_long_trace_lock_2pl():
    from dint_amd.engine import Engine

    p = lt.PARAMS["lock_2pl"]
    eng = Engine(W.TPL, n_slots=p["slots"])
    req, rep = lt.lock_2pl([eng])  # closed loop, 4096-request passes
    _check("lock_2pl", req, rep)
    for batch in (65_536, 1 << 20):  # 65,536: the dominant-slot path of the lock tables; 2^20: one pass per million
        e2 = Engine(W.TPL, n_slots=p["slots"])
        _check("lock_2pl", req, _replay(e2, req, batch))


@pytest.mark.gpu
def test_gpu_long_trace_log_server():
    from dint_amd.engine import Engine

    eng = Engine(W.LOG, log_entries=lt.PARAMS["log_server"]["ring"])
    req, rep = lt.log_server([eng])  # 65,536-request passes
    _check("log_server", req, rep)
    e2 = Engine(W.LOG, log_entries=lt.PARAMS["log_server"]["ring"])
    _check("log_server", req, _replay(e2, req, 4096))
    ring_a, tail_a = eng.read_log(1_000_000)
    ring_b, tail_b = e2.read_log(1_000_000)
    assert tail_a == tail_b == len(req) % 1_000_000 and ring_a.tobytes() == ring_b.tobytes()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_long_trace_store():
    from dint_amd.engine import Engine

    n = lt.PARAMS["store"]["subscribers"]
    for batch in (262_144, 1 << 20):
        eng = Engine(W.STORE, n_rows=n)
        eng.populate(n)
        if batch == 262_144:
            req, rep = lt.store([eng])
        else:
            rep = _replay(eng, req, batch)
        _check("store", req, rep)
        _check_state("store", lt.digest_of_engine("store", eng))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_long_trace_smallbank():
    from dint_amd.engine import Engine

    n = lt.PARAMS["smallbank"]["accounts"]
    engs = []
    for _ in range(3):
        e = Engine(W.SMALLBANK, n_rows=n)
        e.populate(n)
        engs.append(e)
    req, rep = lt.smallbank(engs)  # closed loop: 4096 clients, three servers
    _check("smallbank", req, rep)
    del engs
    e2 = Engine(W.SMALLBANK, n_rows=n)
    e2.populate(n)
    _check("smallbank", req, _replay(e2, req, 1 << 20))
    _check_state("smallbank", lt.digest_of_engine("smallbank", e2))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_long_trace_tatp():
    """the headline workload at the reference's own size: 16,384 restated clients in closed loop against three GPU shard
    servers of 7M subscribers; server 0's 3M requests and replies against the hashes of the unmodified
    tatp/udp/server_shard.cc on the same stream (a single wrong grant sends the clients down another path), then the
    stream again through a fresh engine in passes of 2^20"""
    from dint_amd.engine import Engine

    p = lt.PARAMS["tatp"]

    def mk():
        e = Engine(W.TATP, n_rows=p["subscribers"], log_entries=p["log_entries"])
        e.populate(p["subscribers"])
        return e

    engs = [mk() for _ in range(3)]
    req, rep = lt.tatp(engs)
    _check("tatp", req, rep)
    st = engs[p["server"]].stats()
    assert st["bad_requests"] == 0
    del engs
    e2 = mk()
    _check("tatp", req, _replay(e2, req, 1 << 20))
    _check_state("tatp", lt.digest_of_engine("tatp", e2, lt.n_log_appends("tatp", rep)))
