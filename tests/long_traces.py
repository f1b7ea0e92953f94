"""The long parity traces (tests/golden/long_traces.json): how each one is generated, for the fixture generator and for
the tests alike.  Every generator takes the SERVERS it talks to (objects with submit(ndarray) -> ndarray: CPU oracles
or GPU engines) and returns (requests, replies) of one logical server as the reference would see them.  Test
infrastructure."""
import numpy as np

from dint_amd import wire
from dint_amd.driver import Driver, tpl_trace
from dint_amd.workloads import Zipf

N = 3_000_000
REF_NAME = {"lock_2pl": "lock_2pl", "log_server": "log_server", "store": "store", "smallbank": "smallbank", "tatp": "tatp"}
PARAMS = {
    "lock_2pl": {"slots": 36_000_000, "workers": 4096, "key_space": 24_000_000, "zipf": 0.8, "seed": 2024},
    "log_server": {"ring": 1_000_000, "seed": 2025},
    "store": {"subscribers": 2_000_000, "zipf": 0.8, "p_set": 0.2, "p_missing": 0.03, "seed": 2026},
    "smallbank": {"accounts": 24_000_000, "clients": 4096, "server": 0},
    "tatp": {"subscribers": 7_000_000, "clients": 16_384, "server": 0, "log_entries": 1_000_000},
}


class _Oracle:
    def __init__(self, o):
        self.o = o

    def submit(self, r):
        return self.o.replay(r) if len(r) else r


def fresh_oracle(wl):
    """one CPU oracle server of the workload, populated like the reference (state digests: the closed-loop generators run
    their last epoch past request N, so the state of exactly the first N requests comes from a replay on a fresh server)"""
    from oracle import oracle as orc

    if wl == "store":
        n = PARAMS[wl]["subscribers"]
        return _Oracle(orc.StoreOracle(n * 18 // 4, n))
    if wl == "tatp":
        return _Oracle(orc.TatpOracle(PARAMS[wl]["subscribers"], log_entries=PARAMS[wl]["log_entries"]))
    return _Oracle(orc.SmallbankOracle(PARAMS[wl]["accounts"]))


def oracle_servers(wl):
    from oracle import oracle as orc

    if wl == "lock_2pl":
        return [_Oracle(orc.TplOracle(PARAMS[wl]["slots"]))]
    if wl == "log_server":
        return [_Oracle(orc.LogOracle(PARAMS[wl]["ring"]))]
    if wl == "store":
        n = PARAMS[wl]["subscribers"]
        return [_Oracle(orc.StoreOracle(n * 18 // 4, n))]
    if wl == "tatp":
        return [_Oracle(orc.TatpOracle(PARAMS[wl]["subscribers"], log_entries=PARAMS[wl]["log_entries"])) for _ in range(3)]
    return [_Oracle(orc.SmallbankOracle(PARAMS[wl]["accounts"])) for _ in range(3)]


def lock_2pl(servers):
    p = PARAMS["lock_2pl"]
    req, rep, st = tpl_trace(servers[0], N, n_workers=p["workers"], key_space=p["key_space"], zipf_theta=p["zipf"], seed=p["seed"])
    assert st["protocol_errors"] == 0
    return req, rep


def log_server(servers):
    rng = np.random.default_rng(PARAMS["log_server"]["seed"])
    m = np.zeros(N, wire.LOG_MSG)
    m["key"] = rng.integers(0, 1 << 62, N, dtype=np.uint64)
    m["val"] = rng.integers(0, 256, (N, 40), dtype=np.uint8)
    m["ver"] = rng.integers(0, 1 << 32, N, dtype=np.uint64).astype(np.uint32)
    rep = np.concatenate([servers[0].submit(m[i:i + 65536]) for i in range(0, N, 65536)])
    return m, rep


def store(servers):
    """store/caladan/client_udp.cc:135-147 key shape {s_id, sf_type 1..4, start_time 0/8/16} over the populated
    subscribers, s_id ~ Zipf; READ / SET 80 / 20, 3 % of the keys do not exist"""
    p = PARAMS["store"]
    rng = np.random.default_rng(p["seed"])
    z = Zipf(p["subscribers"], p["zipf"], p["seed"] + 1)
    m = np.zeros(N, wire.STORE_MSG)
    s_id = z.sample(N).astype(np.uint64)
    s_id = np.where(rng.random(N) < p["p_missing"], s_id + np.uint64(p["subscribers"]), s_id)
    m["key"] = s_id | (rng.integers(1, 5, N).astype(np.uint64) << np.uint64(32)) | ((rng.integers(0, 3, N) * 8).astype(np.uint64) << np.uint64(40))
    m["type"] = (rng.random(N) < p["p_set"]).astype(np.uint8)
    m["val"][:, 0], m["val"][:, 1], m["val"][:, 2:6] = rng.integers(0, 24, N), 0x5A, rng.integers(0, 256, (N, 4))
    m["ver"] = rng.integers(0, 1 << 31, N)
    rep = np.concatenate([servers[0].submit(m[i:i + 262144]) for i in range(0, N, 262144)])
    return m, rep


def smallbank(servers):
    """shard server 0's stream of 4096 restated reference clients (the six transactions, the reference's own hot / cold
    account distribution) in closed loop against three servers"""
    p = PARAMS["smallbank"]
    d = Driver(wire.Workload.SMALLBANK, p["clients"], p["accounts"])
    reqs, reps, n = [], [], 0
    while n < N:
        rq = d.next()
        rp = [servers[s].submit(rq[s]) for s in range(3)]
        d.consume(rp)
        reqs.append(rq[p["server"]])
        reps.append(rp[p["server"]])
        n += len(rq[p["server"]])
    return np.concatenate(reqs)[:N], np.concatenate(reps)[:N]


def tatp(servers):
    """shard server 0's stream of 16,384 restated reference clients (the seven transactions in the reference's mix, its own
    subscriber distribution tatp_nurand) in closed loop against three shard servers at the reference's 7M subscribers:
    reads, lock grants and refusals, commits to primary / backups / logs, CALL_FORWARDING inserts and deletes"""
    p = PARAMS["tatp"]
    d = Driver(wire.Workload.TATP, p["clients"], p["subscribers"])
    reqs, reps, n = [], [], 0
    while n < N:
        rq = d.next()
        rp = [servers[s].submit(rq[s]) for s in range(3)]
        d.consume(rp)
        reqs.append(rq[p["server"]])
        reps.append(rp[p["server"]])
        n += len(rq[p["server"]])
    return np.concatenate(reqs)[:N], np.concatenate(reps)[:N]


TRACES = {"lock_2pl": lock_2pl, "log_server": log_server, "store": store, "smallbank": smallbank, "tatp": tatp}


def reply_types(wl, rep):
    f = "action" if wl == "lock_2pl" else "type"
    c = np.bincount(rep[f], minlength=32)
    return {str(k): int(v) for k, v in enumerate(c) if v}


# ---------------------------------------------------------------------------------------------------------------
# Final-state digests of the kv workloads: one canonical byte stream per workload, built the same way
# from the unmodified reference's dump file (oracle/ref_harness/*: rows in bucket / chain order, lock words, log ring), from
# the CPU oracle and from a GPU engine, and hashed.  Canonical = what the reference defines: the value bytes of a row
# nobody wrote since the populate (ver == 0) are cut down to the bytes the populate assigns (the reference copies
# partially initialised stack structs there, tatp/udp/tatp.h:291-307), and a DELETE_LOG record's val bytes are dropped
# (the reference leaves them as they were, tatp/udp/server_shard.cc:196-207).
import hashlib

VS = {"store": 40, "tatp": 40, "smallbank": 8}
N_TABLES = {"store": 1, "tatp": 5, "smallbank": 2}


def _rows_canon(wl, t, keys, vers, vals):
    from oracle import oracle as orc

    vals = np.array(vals, "u1", copy=True).reshape(len(keys), VS[wl])
    cols = orc.STORE_ASSIGNED if wl == "store" else orc.TATP_ASSIGNED.get(t) if wl == "tatp" else None
    if cols is not None:
        keep = np.zeros(VS[wl], bool)
        keep[cols] = True
        vals[np.ix_(np.asarray(vers) == 0, ~keep)] = 0
    rec = np.zeros(len(keys), np.dtype([("key", "<u8"), ("ver", "<u4"), ("val", "u1", (VS[wl],))]))
    rec["key"], rec["ver"], rec["val"] = keys, vers, vals
    return rec.tobytes()


def _log_canon(tail, recs):
    recs = np.array(recs, "u1", copy=True).reshape(-1, 64)
    recs[recs[:, 52] == 1, 8:48] = 0  # DELETE_LOG
    recs[:, 54:] = 0
    return np.array([tail, len(recs)], "<u4").tobytes() + recs.tobytes()


def digest_of_reference_dump(wl, dump: bytes) -> str:
    """the state dump the replay harness writes after the trace (ref_harness/kvs_dump.h, ref_tatp.cc, ref_smallbank.cc)"""
    h, off, row = hashlib.sha256(), 0, 12 + VS[wl]
    for t in range(N_TABLES[wl]):
        n = int(np.frombuffer(dump, "<u8", 1, off)[0]); off += 8
        r = np.frombuffer(dump, np.dtype([("key", "<u8"), ("ver", "<u4"), ("val", "u1", (VS[wl],))]), n, off); off += n * row
        h.update(_rows_canon(wl, t, r["key"], r["ver"], r["val"]))
    if wl == "tatp":
        for t in range(5):
            cnt = int(np.frombuffer(dump, "<u4", 1, off)[0]); off += 4
            h.update(np.frombuffer(dump, "<u4", cnt, off).tobytes()); off += 4 * cnt
        tail, n = (int(x) for x in np.frombuffer(dump, "<u4", 2, off)); off += 8
        h.update(_log_canon(tail, np.frombuffer(dump, "u1", n * 64, off))); off += n * 64
    elif wl == "smallbank":
        for t in range(2):
            cnt = int(np.frombuffer(dump, "<u4", 1, off)[0]); off += 4
            h.update(np.frombuffer(dump, "<u4", 3 * cnt, off).tobytes()); off += 12 * cnt
    assert off == len(dump), (off, len(dump))
    return h.hexdigest()


def digest_of_oracle(wl, o, n_log: int = 0) -> str:
    h = hashlib.sha256()
    for t in range(N_TABLES[wl]):
        k, v, x = o.dump() if wl == "store" else o.dump(t)
        h.update(_rows_canon(wl, t, k, v, x))
    if wl == "tatp":
        for t in range(5):
            h.update(np.nonzero(o.locks(t))[0].astype("<u4").tobytes())
        h.update(_log_canon(o.tail, o.ring[:min(n_log, o.cap)]))
    elif wl == "smallbank":
        for t in range(2):
            ex, sh = o.num_ex(t), o.num_sh(t)
            nz = np.nonzero(ex | sh)[0]
            h.update(np.stack([nz.astype("<u4"), ex[nz], sh[nz]], 1).astype("<u4").tobytes())
    return h.hexdigest()


def digest_of_engine(wl, eng, n_log: int = 0) -> str:
    h = hashlib.sha256()
    for t in range(N_TABLES[wl]):
        k, v, x = eng.dump_rows(t)
        h.update(_rows_canon(wl, t, k, v, x))
    if wl == "tatp":
        for t in range(5):
            a, _ = eng.read_locks(t)
            h.update(np.nonzero(a)[0].astype("<u4").tobytes())
        cap = PARAMS["tatp"]["log_entries"]
        ring, tail = eng.read_log(cap)
        h.update(_log_canon(tail, np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64)[:min(n_log, cap)]))
    elif wl == "smallbank":
        for t in range(2):
            ex, sh = eng.read_locks(t)
            nz = np.nonzero(ex | sh)[0]
            h.update(np.stack([nz.astype("<u4"), ex[nz], sh[nz]], 1).astype("<u4").tobytes())
    return h.hexdigest()


def n_log_appends(wl, rep) -> int:
    """log records the trace appended (= replies of the log ack types): how much of the ring belongs to the state"""
    if wl == "tatp":
        return int(np.isin(rep["type"], (17, 27)).sum())
    return 0
