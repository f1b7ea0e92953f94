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
