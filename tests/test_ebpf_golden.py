"""Parity with DINT proper -- the reference's eBPF servers.

tests/golden/ebpf_*.npz hold request streams and the replies of the UNMODIFIED <wl>/ebpf/*_kern.c (XDP + TC programs)
and *_user.c (user-space fallback), run under the emulator of oracle/ref_harness/ebpf (tests/golden/make_golden_ebpf.py;
the programs cannot be loaded here, but they are plain C and compile with the host gcc against a stub of
bpf_helpers.h).  They pin the codes only the eBPF flavour has -- REJECT_LOCK_SAME_KEY (tatp/ebpf/lock_kern.c:289-298),
WARMUP_READ (smallbank/ebpf/shard_kern.c:585-667, shard_user.c:179-186), store INSERT (store/ebpf/store_kern.c:226-297)
-- and the eBPF twin of every server on client-shaped traffic.  CPU: the oracle against the fixtures; GPU: the engines.
Reply bytes where the two reference flavours themselves differ are masked as the fixtures say (`skip_*`)."""
import json
import os

import numpy as np
import pytest

from dint_amd import wire
from oracle import oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
W = wire.Workload


def _load(name, dtype, req="req", rep="rep"):
    z = np.load(os.path.join(G, name + ".npz"))
    return z, json.loads(str(z["meta"])), np.frombuffer(z[req].tobytes(), dtype), np.frombuffer(z[rep].tobytes(), dtype)


def _mask(z, key, n):
    return np.unpackbits(z[key])[:n].astype(bool)


def _same_store(got, rep, z):
    a, b = got.copy(), rep.copy()
    m = _mask(z, "skip_ver", len(rep))
    a["ver"][m], b["ver"][m] = 0, 0
    return a.tobytes() == b.tobytes()


def _same_tatp(got, rep, z):
    a, b = got.copy(), rep.copy()
    m = _mask(z, "skip_val", len(rep))
    a["val"][m], b["val"][m] = 0, 0
    return a.tobytes() == b.tobytes()


# ------------------------------------------------------------------------------------------------- CPU: the oracle
def test_ebpf_micro_servers_equal_the_udp_fixtures():
    z = np.load(os.path.join(G, "ebpf_micro.npz"))
    for name in ("lock_fasst", "log_server"):  # the eBPF servers' replies to the udp fixtures' requests
        assert z[name + "_rep"].tobytes() == np.load(os.path.join(G, name + ".npz"))["rep"].tobytes()


def test_oracle_vs_ebpf_lock_2pl():
    z, meta, req, rep = _load("ebpf_micro", wire.TPL_MSG, "tpl_req", "tpl_rep")
    got = orc.TplOracle(meta["nslots"]).replay(req)
    assert got.tobytes() == rep.tobytes()
    assert (rep["action"] == wire.Tpl.REJECT_LOCK).sum() > 1000 and (rep["action"] == wire.Tpl.GRANT_LOCK).sum() > 5000


def test_oracle_vs_ebpf_tatp_lock_same_key():
    z, meta, req, rep = _load("ebpf_tatp_lock", wire.TATP_MSG)
    o = orc.TatpOracle(meta["n_sub"], populate_n=0)
    o.same_key_mode()
    assert o.replay(req).tobytes() == rep.tobytes()
    c = np.bincount(rep["type"], minlength=29)
    assert c[wire.Tatp.REJECT_LOCK] > 300 and c[28] > 1000 and c[wire.Tatp.GRANT_LOCK] > 1000


def test_oracle_vs_ebpf_store_insert():
    z, meta, req, rep = _load("ebpf_store", wire.STORE_MSG)
    got = orc.StoreOracle(meta["buckets"], 0).replay(req)
    assert _same_store(got, rep, z)
    assert (rep["type"][req["type"] == 2] == wire.Store.INSERT_ACK).all() and meta["user_path"] > 100


@pytest.mark.slow
def test_oracle_vs_ebpf_smallbank_warmup_and_clients():
    z, meta, req, rep = _load("ebpf_smallbank", wire.SB_MSG)
    got = orc.SmallbankOracle(meta["n_acct"]).replay(req)
    assert got.tobytes() == rep.tobytes()
    assert (rep["type"][:meta["warmups"]] == 18).all() and (rep["val"][:meta["warmups"], 0] >= 97).all()


def test_oracle_vs_ebpf_tatp_clients():
    z, meta, req, rep = _load("ebpf_tatp", wire.TATP_MSG)
    got = orc.TatpOracle(meta["n_sub"], populate_n=0).replay(req)
    assert _same_tatp(got, rep, z)
    assert meta["populate_inserts"] > 5000 and (rep["type"][:meta["populate_inserts"]] == wire.Tatp.INSERT_PRIM_ACK).all()


@pytest.mark.ref
def test_fixtures_still_equal_the_emulated_servers():
    """where the reference tree and the emulator binaries exist (this container): the fixtures are what they produce"""
    if not orc.ebpf_available("tatp_lock"):
        pytest.skip("oracle/_ref/ref_ebpf_* not built (no /root/reference here)")
    for name, wl, dt, rq, rp in (("ebpf_micro", "lock_2pl", wire.TPL_MSG, "tpl_req", "tpl_rep"),
                                 ("ebpf_tatp_lock", "tatp_lock", wire.TATP_MSG, "req", "rep"),
                                 ("ebpf_tatp", "tatp", wire.TATP_MSG, "req", "rep")):
        z, meta, req, rep = _load(name, dt, rq, rp)
        got, st = orc.ebpf_replay(wl, req)
        assert got.tobytes() == rep.tobytes() and st["unanswered"] == 0, name


# ------------------------------------------------------------------------------------------------- GPU: the engines
@pytest.mark.gpu
def test_gpu_vs_ebpf_lock_2pl():
    from dint_amd.engine import Engine

    z, meta, req, rep = _load("ebpf_micro", wire.TPL_MSG, "tpl_req", "tpl_rep")
    e = Engine(W.TPL, n_slots=meta["nslots"])
    assert e.submit(req).tobytes() == rep.tobytes()


@pytest.mark.gpu
def test_gpu_vs_ebpf_tatp_lock_same_key():
    from dint_amd.engine import Engine

    z, meta, req, rep = _load("ebpf_tatp_lock", wire.TATP_MSG)
    e = Engine(W.TATP, n_rows=meta["n_sub"], flags=4)  # DINT_FLAG_LOCK_SAME_KEY
    assert e.submit(req).tobytes() == rep.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 1])
def test_gpu_vs_ebpf_store_insert(flags):
    from dint_amd.engine import Engine

    z, meta, req, rep = _load("ebpf_store", wire.STORE_MSG)
    e = Engine(W.STORE, n_rows=meta["buckets"] * 4 // 18, flags=flags)
    assert e.hash_size(0) == meta["buckets"]
    assert _same_store(e.submit(req), rep, z)


@pytest.mark.gpu
def test_gpu_vs_ebpf_smallbank_warmup_and_clients():
    from dint_amd.engine import Engine

    z, meta, req, rep = _load("ebpf_smallbank", wire.SB_MSG)
    e = Engine(W.SMALLBANK, n_rows=meta["n_acct"])
    e.populate(meta["n_acct"])
    assert e.submit(req).tobytes() == rep.tobytes()


@pytest.mark.gpu
def test_gpu_vs_ebpf_tatp_clients():
    from dint_amd.engine import Engine

    z, meta, req, rep = _load("ebpf_tatp", wire.TATP_MSG)
    e = Engine(W.TATP, n_rows=meta["n_sub"])  # the eBPF flavour starts empty: the trace's INSERT_PRIMs populate it
    assert _same_tatp(e.submit(req), rep, z)
    assert e.stats()["missing_keys"] == 0
