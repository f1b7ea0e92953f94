#!/usr/bin/env python3
"""Generate tests/golden/ebpf_*.npz by running the UNMODIFIED reference eBPF servers -- DINT proper: the XDP and TC
programs of <wl>/ebpf/*_kern.c and the user-space fallback of *_user.c -- under the emulator of
oracle/ref_harness/ebpf (`make -C oracle ref_ebpf`: host gcc against stubs of bpf_helpers.h / libbpf; the programs
cannot be loaded here, but they are plain C).  Only runs where /root/reference exists; the fixtures are committed.

    python tests/golden/make_golden_ebpf.py [name ...]

These are the pins of the codes only the eBPF flavour has -- REJECT_LOCK_SAME_KEY
(tatp/ebpf/lock_kern.c:289-298), WARMUP_READ (smallbank/ebpf/shard_kern.c:585-667 + shard_user.c:179-186), store INSERT
(store/ebpf/store_kern.c:226-297) -- and of the eBPF twin of every udp/ server on client-shaped traffic.  Where the two
flavours differ in a serial replay the fixture says so in `meta` and carries a mask (`skip`: reply bytes not compared):
  * lock_2pl / smallbank: the eBPF counters are signed and tested `> 0` (lock_2pl/ebpf/ls_kern.c:67-86), the udp ones
    unsigned and tested `== 0` (lock_2pl/udp/server.cc:84-106): they part ways after a RELEASE nobody holds, which no
    client sends -- these traces release only what was granted;
  * store: a SET that misses the in-kernel cache is answered by user space with the NEW version in `ver`
    (store/ebpf/store_user.c:146-152), and a NOT_EXIST from user space carries the eviction flag there; udp echoes `ver`;
  * tatp: a DELETE_*_ACK carries the bucket's recomputed bloom filter in val[0..8] (tatp/ebpf/shard_user.c:205-207).
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import tracegen  # noqa: E402
from dint_amd import wire  # noqa: E402
from dint_amd.driver import Driver  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from shard_double import fasthash_key, fasthash_lid  # noqa: E402

N_SLOTS, STORE_BUCKETS, TATP_SUB, SB_ACCT = 36_000_000, 9_000_000, 7_000_000, 24_000_000


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")


def raw(a):
    return np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint8)


def replay(wl, req):
    rep, st = orc.ebpf_replay(wl, req)
    assert st["unanswered"] == 0, st
    return rep, st


def gen_micro():
    """lock_fasst and log_server: the udp fixtures' own request streams (the two flavours must agree byte for byte);
    lock_2pl: a well-formed trace -- 512 closed-loop holders over 200 hot + 24M lids: acquire, release what was granted"""
    out = {}
    for name, dt in (("lock_fasst", wire.FASST_MSG), ("log_server", wire.LOG_MSG)):
        g = np.load(os.path.join(HERE, name + ".npz"))
        req = np.frombuffer(g["req"].tobytes(), dt)
        rep, st = replay(name, req)
        assert rep.tobytes() == g["rep"].tobytes(), f"{name}: the eBPF and udp servers disagree on the udp fixture"
        out[name + "_rep"] = raw(rep)
    rng = np.random.default_rng(21)
    o = orc.TplOracle(N_SLOTS)
    held, reqs = [], []
    for _ in range(30000):
        m = np.zeros(1, wire.TPL_MSG)
        if held and rng.random() < 0.45:
            lid, ty = held.pop(int(rng.integers(0, len(held))))
            m["action"], m["lid"], m["type"] = 1, lid, ty
            o.replay(m)
        else:
            lid = int(rng.integers(0, 200)) if rng.random() < 0.7 else int(rng.integers(0, 24_000_000))
            ty = int(rng.random() < 0.4)
            m["action"], m["lid"], m["type"] = 0, lid, ty
            if o.replay(m)["action"][0] == wire.Tpl.GRANT_LOCK:
                held.append((lid, ty))
        reqs.append(m)
    req = np.concatenate(reqs)
    rep, st = replay("lock_2pl", req)
    urep, _ = orc.ref_replay("lock_2pl", req)
    assert rep.tobytes() == urep.tobytes(), "lock_2pl: eBPF and udp disagree on a well-formed trace"
    save("ebpf_micro", tpl_req=raw(req), tpl_rep=raw(rep), **out,
         meta=json.dumps({"nslots": N_SLOTS, "fasst_log": "replies of the eBPF servers to the requests of lock_fasst.npz / log_server.npz",
                          "tpl": "30000 requests, releases only of granted locks"}))


def _aliases(table_buckets, n_pairs, rng):
    """pairs of keys whose tatp lock slots coincide: lock_hash = fasthash64(key) % (4 * hash_size), tatp/ebpf/lock_kern.c:262"""
    keys = rng.choice(1 << 40, 3_000_000, replace=False).astype(np.uint64)
    slot = fasthash_key(keys) % np.uint64(4 * table_buckets)
    order = np.argsort(slot, kind="stable")
    s = slot[order]
    dup = np.nonzero(s[1:] == s[:-1])[0][:n_pairs]
    return keys[order[dup]], keys[order[dup + 1]]


def gen_tatp_lock():
    """tatp/ebpf/lock_kern.c: ACQUIRE_LOCK / ABORT on tables 0..3 (the call-forwarding table has another size in the
    eBPF build -- 15/4 against udp's 45/8 rows per subscriber, tatp/ebpf/utils.h:21 -- so its slots alias differently),
    300 keys per table of which 40 pairs share a lock slot: GRANT / REJECT_LOCK / REJECT_LOCK_SAME_KEY all occur"""
    T = wire.Tatp
    rng = np.random.default_rng(22)
    hs = [TATP_SUB * 3 // 2 // 4] * 2 + [TATP_SUB * 15 // 4 // 4] * 2
    pools = []
    for t in range(4):
        a, b = _aliases(hs[t], 40, rng)
        pools.append(np.concatenate([a, b, rng.choice(1 << 40, 220, replace=False).astype(np.uint64)]))
    n = 60000
    m = np.zeros(n, wire.TATP_MSG)
    m["ord"] = rng.integers(0, 256, n)
    m["table"] = rng.integers(0, 4, n)
    m["key"] = [pools[t][rng.integers(0, len(pools[t]))] for t in m["table"]]
    m["type"] = np.where(rng.random(n) < 0.6, T.ACQUIRE_LOCK, T.ABORT)
    m["val"], m["ver"] = 0xAB, rng.integers(0, 4, n)  # echoed; kept compressible (the fixture is committed)
    rep, st = replay("tatp_lock", m)
    c = np.bincount(rep["type"], minlength=29)
    assert c[T.GRANT_LOCK] > 1000 and c[T.REJECT_LOCK] > 300 and c[28] > 1000, c
    save("ebpf_tatp_lock", req=raw(m), rep=raw(rep), meta=json.dumps({"n_sub": TATP_SUB, "flag": "DINT_FLAG_LOCK_SAME_KEY"}))


def _client_stream(workload, servers, n_clients, n_rows, epochs):
    """shard server 0's request stream of the restated reference clients (closed loop against three CPU oracles)"""
    d = Driver(workload, n_clients, n_rows)
    reqs = []
    for _ in range(epochs):
        rq = d.next()
        d.consume([servers[s].replay(rq[s]) if len(rq[s]) else rq[s] for s in range(3)])
        reqs.append(rq[0])
    return np.concatenate(reqs), d.stats()


def gen_smallbank():
    """smallbank/ebpf: 2 x 1500 WARMUP_READs of cold accounts (first touch: user space answers WARMUP_READ_ACK with the
    row, shard_user.c:179-186), then 40 epochs of the six smallbank transactions from 1000 clients"""
    rng = np.random.default_rng(23)
    acct = rng.choice(SB_ACCT, 1500, replace=False).astype(np.uint64)
    w = np.zeros(2 * len(acct), wire.SB_MSG)
    w["type"], w["table"], w["key"], w["ver"], w["val"] = 17, np.repeat([0, 1], len(acct)), np.tile(acct, 2), 0x55, 0xEE
    srv = [orc.SmallbankOracle(SB_ACCT) for _ in range(3)]
    traffic, st = _client_stream(wire.Workload.SMALLBANK, srv, 1000, SB_ACCT, 40)
    req = np.concatenate([w, traffic])
    rep, est = replay("smallbank", req)
    assert (rep["type"][:len(w)] == 18).all()
    save("ebpf_smallbank", req=raw(req), rep=raw(rep),
         meta=json.dumps({"n_acct": SB_ACCT, "warmups": len(w), "client_txns": int(st["txns"]), "user_path": est["pass"]}))


def gen_store():
    """store/ebpf: ~20300 INSERTs (of which ~350 keys crowd 32 buckets: evictions, the user-space path), then 60000
    READ / SET over them and over keys that do not exist"""
    rng = np.random.default_rng(24)
    crowd = []  # keys of buckets 0..31: ~10 each out of 100M consecutive integers (a bucket's cache line holds 4)
    for lo in range(0, 100_000_000, 10_000_000):
        blk = np.arange(1 << 41, (1 << 41) + 10_000_000, dtype=np.uint64) + np.uint64(lo)
        crowd.append(blk[fasthash_key(blk) % np.uint64(STORE_BUCKETS) < 32])
    crowd = np.concatenate(crowd)
    keys = np.unique(np.concatenate([crowd, rng.choice(1 << 40, 20000, replace=False).astype(np.uint64)]))
    rng.shuffle(keys)
    nk = len(keys)
    ins = np.zeros(nk, wire.STORE_MSG)
    ins["type"], ins["key"], ins["ver"] = 2, keys, rng.integers(0, 4, nk)
    ins["val"][:, 0], ins["val"][:, 1] = rng.integers(0, 24, nk), 0x5A  # {end_time, magic} as store/caladan/client_ebpf.cc
    n = 60000
    tr = np.zeros(n, wire.STORE_MSG)
    tr["key"] = np.where(rng.random(n) < 0.95, keys[rng.integers(0, nk, n)], rng.integers(1 << 42, 1 << 43, n).astype(np.uint64))
    tr["type"] = (rng.random(n) < 0.35).astype(np.uint8)
    tr["val"][:, 0], tr["val"][:, 1], tr["val"][:, 2], tr["ver"] = rng.integers(0, 24, n), 0x5A, rng.integers(0, 256, n), rng.integers(0, 4, n)
    req = np.concatenate([ins, tr])
    rep, st = replay("store", req)
    # `ver` of a reply that came from user space is not the udp flavour's (see the module docstring): found by comparing
    # with the udp-semantics oracle, recorded as a mask
    want = orc.StoreOracle(STORE_BUCKETS, 0).replay(req)
    a, w = rep.copy(), want.copy()
    a["ver"], w["ver"] = 0, 0
    assert a.tobytes() == w.tobytes(), "store: eBPF and udp disagree beyond the ver field"
    skip = rep["ver"] != want["ver"]
    assert st["pass"] > 100 and skip.sum() < st["pass"], (st, int(skip.sum()))
    assert not skip[req["type"] == 2].any()
    save("ebpf_store", req=raw(req), rep=raw(rep), skip_ver=np.packbits(skip),
         meta=json.dumps({"buckets": STORE_BUCKETS, "inserts": nk, "user_path": st["pass"], "ver_masked": int(skip.sum())}))


def gen_tatp():
    """tatp/ebpf/shard_kern.c: the eBPF flavour starts empty and is populated by the clients' INSERT_PRIMs
    (tatp/caladan/client_ebpf_shard.cc:96-339): the rows of 1000 subscribers, then 40 epochs of the seven tatp
    transactions from 800 clients"""
    T = wire.Tatp
    NS = 1000
    srv = [orc.TatpOracle(TATP_SUB, populate_n=NS) for _ in range(3)]
    pop = []
    for t in range(5):
        k, _, vals = srv[0].dump(t)
        m = np.zeros(len(k), wire.TATP_MSG)
        m["type"], m["table"], m["key"], m["val"] = T.INSERT_PRIM, t, k, vals
        pop.append(m)
    pop = np.concatenate(pop)
    traffic, st = _client_stream(wire.Workload.TATP, srv, 800, NS, 40)
    req = np.concatenate([pop, traffic])
    rep, est = replay("tatp", req)
    isdel = np.isin(req["type"], [T.DELETE_PRIM, T.DELETE_BCK])
    save("ebpf_tatp", req=raw(req), rep=raw(rep), skip_val=np.packbits(isdel),
         meta=json.dumps({"n_sub": TATP_SUB, "populate_inserts": len(pop), "client_txns": int(st["txns"]),
                          "delete_acks_val_masked": int(isdel.sum()), "user_path": est["pass"]}))


def gen_backpressure():
    """The replies of a CONTENDED entry: REJECT_READ / REJECT_COMMIT / REJECT_SET / REJECT_INSERT /
    RETRY come from a failed CAS on the spin lock of a cache entry or lock unit (tatp/ebpf/shard_kern.c:173-178,371-376,
    store/ebpf/store_kern.c:57-66,140-150,226-236, lock_2pl/ebpf/ls_kern.c:59-64, smallbank/ebpf/shard_kern.c:122-152):
    another packet holds it.  A serial replay never fails that CAS -- so the emulator, which owns the map memory, holds
    the lock words itself during the request (EMU_HOLD) and records what the UNMODIFIED programs answer.  Every request
    type of every server, on existing and on missing keys; dint_refuse / rt_refuse are held to these bytes."""
    out, meta = {}, {}
    rng = np.random.default_rng(77)

    def stream(dt, types, keyf, n_each=24, table=None):
        m = np.zeros(len(types) * n_each, dt)
        m["type" if "type" in dt.names and dt is not wire.TPL_MSG else "action"] = np.repeat(types, n_each)
        keyf(m)
        if "val" in dt.names:
            m["val"] = rng.integers(0, 256, m["val"].shape, dtype=np.uint8)
        if "ver" in dt.names:
            m["ver"] = rng.integers(0, 1 << 31, len(m))
        if table is not None:
            m["table"] = rng.integers(0, table, len(m))
        return m

    T, S, B = wire.Tatp, wire.Store, wire.Sb
    cases = {
        "lock_2pl": (stream(wire.TPL_MSG, [0, 1], lambda m: (m.__setitem__("lid", rng.integers(0, 1000, len(m))),
                                                            m.__setitem__("type", rng.integers(0, 2, len(m))))), 2),
        "store": (stream(wire.STORE_MSG, [S.READ, S.SET, S.INSERT], lambda m: m.__setitem__("key", rng.integers(0, 1 << 40, len(m)))), 1),
        "tatp": (stream(wire.TATP_MSG, [T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.COMMIT_BCK, T.COMMIT_LOG, T.INSERT_PRIM,
                                        T.INSERT_BCK, T.DELETE_PRIM, T.DELETE_BCK, T.DELETE_LOG],
                        lambda m: m.__setitem__("key", rng.integers(0, 1 << 40, len(m))), table=5), 1),
        "smallbank": (stream(wire.SB_MSG, [0, 1, 2, 3, 4, 5, 6, 17], lambda m: m.__setitem__("key", rng.integers(0, 5000, len(m))), table=2), 3),
    }
    for wl, (req, hold) in cases.items():
        free, _ = orc.ebpf_replay(wl, req)               # the same requests with nothing held: the serial answers
        rep, st = orc.ebpf_replay(wl, req, hold=hold)    # ... and with the entries' locks held by "another packet"
        f = "action" if wl == "lock_2pl" else "type"
        out[wl + "_req"], out[wl + "_rep"], out[wl + "_free"] = raw(req), raw(rep), raw(free)
        meta[wl] = {"hold_bits": hold, "n": len(req),
                    "held_reply_types": {str(k): int(v) for k, v in enumerate(np.bincount(rep[f], minlength=32)) if v},
                    "free_reply_types": {str(k): int(v) for k, v in enumerate(np.bincount(free[f], minlength=32)) if v}}
        print(wl, meta[wl])
    save("ebpf_backpressure", meta=json.dumps(meta), **out)


GEN = {"backpressure": gen_backpressure, "micro": gen_micro, "tatp_lock": gen_tatp_lock, "smallbank": gen_smallbank, "store": gen_store, "tatp": gen_tatp}

if __name__ == "__main__":
    orc.build()
    for w in sys.argv[1:] or list(GEN):
        t = time.time()
        GEN[w]()
        print(f"{w}: {time.time()-t:.1f}s")
