"""Seeded random request traces for the parity tests (test tooling, numpy only).

The traces are adversarial rather than realistic: a small hot key set makes most
requests of a batch collide on a few slots/buckets, so in-batch ordering matters.
"""
from __future__ import annotations

import numpy as np

from dint_amd import wire


def fasst_random(n, seed=0, n_hot=64, key_space=24_000_000, p_hot=0.7):
    rng = np.random.default_rng(seed)
    m = np.zeros(n, wire.FASST_MSG)
    m["type"] = rng.integers(0, 4, n)
    hot = rng.integers(0, 2**32, n_hot, dtype=np.uint64).astype("<u4")
    cold = rng.integers(0, key_space, n).astype("<u4")
    m["lid"] = np.where(rng.random(n) < p_hot, hot[rng.integers(0, n_hot, n)], cold)
    m["ver"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype("<u4")
    return m


def tpl_random(n, seed=0, n_hot=64, key_space=24_000_000, p_hot=0.7):
    rng = np.random.default_rng(seed)
    m = np.zeros(n, wire.TPL_MSG)
    m["action"] = rng.integers(0, 2, n)
    m["type"] = rng.integers(0, 2, n)
    hot = rng.integers(0, 2**32, n_hot, dtype=np.uint64).astype("<u4")
    cold = rng.integers(0, key_space, n).astype("<u4")
    m["lid"] = np.where(rng.random(n) < p_hot, hot[rng.integers(0, n_hot, n)], cold)
    return m


def log_random(n, seed=0):
    rng = np.random.default_rng(seed)
    m = np.zeros(n, wire.LOG_MSG)
    m["type"] = wire.Log.COMMIT
    m["key"] = rng.integers(0, 2**63, n, dtype=np.uint64)
    m["val"] = rng.integers(0, 256, (n, 40), dtype=np.uint8)
    m["ver"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype("<u4")
    return m


def store_key(s_id, sf_type, start_time):
    return (np.asarray(s_id, np.uint64) | (np.asarray(sf_type, np.uint64) << np.uint64(32))
            | (np.asarray(start_time, np.uint64) << np.uint64(40)))


def store_random(n, seed=0, n_sub_touch=50, p_set=0.4, p_missing=0.1, p_insert=0.0):
    """READ/SET (optionally INSERT of fresh keys) over the rows of the first
    `n_sub_touch` subscribers of a store populated per store/udp/tatp.h:44-66."""
    rng = np.random.default_rng(seed)
    m = np.zeros(n, wire.STORE_MSG)
    s_id = rng.integers(0, n_sub_touch, n)
    sf = rng.integers(1, 5, n)
    st = rng.integers(0, 3, n) * 8
    key = store_key(s_id, sf, st)
    missing = rng.random(n) < p_missing
    key = np.where(missing, key | (np.uint64(0xDEAD) << np.uint64(48)), key)
    m["key"] = key
    u = rng.random(n)
    m["type"] = np.where(u < p_set, wire.Store.SET, wire.Store.READ)
    m["val"] = rng.integers(0, 256, (n, 40), dtype=np.uint8)
    m["ver"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype("<u4")
    if p_insert > 0:
        ins = rng.random(n) < p_insert
        # fresh keys: s_id far above any populated subscriber, unique per request
        fresh = store_key(3_000_000_000 + np.arange(n), 1, 0)
        m["key"] = np.where(ins, fresh, m["key"])
        m["type"] = np.where(ins, wire.Store.INSERT, m["type"])
        # later READ/SET of an inserted key
        later = np.nonzero(~ins)[0]
        pick = later[rng.random(len(later)) < p_insert * 2]
        src = rng.integers(0, n, len(pick))
        ok = ins[src] & (src < pick)
        m["key"][pick[ok]] = fresh[src[ok]]
    return m


# --- tatp ------------------------------------------------------------------ #
T = wire.Tatp


def tatp_random(n, existing, seed=0, n_sub_touch=40, well_formed=True):
    """Random mix of all 13 tatp request types.

    `existing` = list of 5 key arrays (rows that exist initially, e.g. from the
    oracle's populate dump restricted to s_id < n_sub_touch).  With
    well_formed=True the generator tracks existence so that COMMIT/DELETE only
    hit existing rows and INSERT only missing ones (the cases on which the
    reference udp server does not panic, tatp/udp/kvs.h:91,152).
    """
    rng = np.random.default_rng(seed)
    m = np.zeros(n, wire.TATP_MSG)
    live = [set(int(k) for k in ks) for ks in existing]
    pools = []
    for t in range(5):
        pool = set(live[t])
        # add plausible missing keys
        for s in range(n_sub_touch):
            if t == 0:
                pool.add(s)
            elif t in (2, 3):
                for a in range(1, 5):
                    pool.add(s | (a << 32))
            elif t == 4:
                for a in range(1, 5):
                    for st in (0, 8, 16):
                        pool.add(s | (a << 32) | (st << 40))
        pools.append(sorted(pool))
    types = [T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.COMMIT_BCK, T.COMMIT_LOG, T.INSERT_PRIM,
             T.INSERT_BCK, T.DELETE_PRIM, T.DELETE_BCK, T.DELETE_LOG]
    weights = np.array([30, 12, 8, 8, 6, 6, 7, 5, 7, 5, 6], float)
    weights /= weights.sum()
    ty = rng.choice(len(types), n, p=weights)
    tb = rng.choice(5, n, p=[0.2, 0.1, 0.2, 0.2, 0.3])
    u = rng.integers(0, 2**31, n)
    vals = rng.integers(0, 256, (n, 40), dtype=np.uint8)
    m["ord"] = rng.integers(0, 256, n)
    m["ver"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype("<u4")
    m["val"] = vals
    for i in range(n):
        t = int(tb[i])
        op = types[ty[i]]
        pool = pools[t]
        key = pool[u[i] % len(pool)]
        if well_formed:
            ex = key in live[t]
            if op in (T.COMMIT_PRIM, T.COMMIT_BCK, T.DELETE_PRIM, T.DELETE_BCK) and not ex:
                op = T.INSERT_PRIM if op in (T.COMMIT_PRIM, T.DELETE_PRIM) else T.INSERT_BCK
            elif op in (T.INSERT_PRIM, T.INSERT_BCK) and ex:
                op = T.COMMIT_PRIM if op == T.INSERT_PRIM else T.COMMIT_BCK
        if op in (T.INSERT_PRIM, T.INSERT_BCK):
            live[t].add(key)
        elif op in (T.DELETE_PRIM, T.DELETE_BCK):
            live[t].discard(key)
        m["type"][i] = op
        m["table"][i] = t
        m["key"][i] = key
    return m


# --- smallbank --------------------------------------------------------------- #
def sb_random(n, seed=0, n_acct_touch=40):
    rng = np.random.default_rng(seed)
    m = np.zeros(n, wire.SB_MSG)
    m["ord"] = rng.integers(0, 256, n)
    m["type"] = rng.choice(7, n, p=[0.22, 0.18, 0.16, 0.12, 0.12, 0.1, 0.1])
    m["table"] = rng.integers(0, 2, n)
    m["key"] = rng.integers(0, n_acct_touch, n)
    m["val"] = rng.integers(0, 256, (n, 8), dtype=np.uint8)
    m["ver"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype("<u4")
    return m


# --- masks for the reference's uninitialised populate bytes ------------------ #
from oracle.oracle import STORE_ASSIGNED, TATP_ASSIGNED, mask_populate_garbage  # noqa: E402,F401  (shared with bench.py's cpu_baseline leg)
