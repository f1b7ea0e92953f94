#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the UNMODIFIED reference
udp/ servers (oracle/_ref/ref_*, built by `make -C oracle ref` from /root/reference)
over seeded random traces.  Only runs where /root/reference exists; the fixtures it
writes are committed so the GPU box and CI never need the reference tree.

    python tests/golden/make_golden.py [workload ...]

Each fixture <workload>.npz holds: req (raw request bytes), rep (raw reply bytes of the
reference), and for the lock tables / logs the reference's final state dump.
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
from oracle import oracle as orc  # noqa: E402

# reference sizes (compile-time constants of the reference)
N_SLOTS = 36_000_000        # lock_fasst/udp/utils.h:12, lock_2pl/udp/utils.h
STORE_SUB = 2_000_000       # store/udp/tatp.h:10
TATP_SUB = 7_000_000        # tatp/udp/tatp.h:28
SB_ACCT = 24_000_000        # smallbank/udp/smallbank.h:17


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")


def raw(a):
    return np.frombuffer(a.tobytes(), np.uint8)


def gen_lock_fasst():
    req = tracegen.fasst_random(30000, seed=11)
    rep, st, dump = orc.ref_replay("lock_fasst", req, dump=True)
    save("lock_fasst", req=raw(req), rep=raw(rep), dump=np.frombuffer(dump, np.uint8), meta=json.dumps({"nslots": N_SLOTS}))


def gen_lock_2pl():
    req = tracegen.tpl_random(30000, seed=12)
    rep, st, dump = orc.ref_replay("lock_2pl", req, dump=True)
    save("lock_2pl", req=raw(req), rep=raw(rep), dump=np.frombuffer(dump, np.uint8), meta=json.dumps({"nslots": N_SLOTS}))


def gen_log_server():
    req = tracegen.log_random(5000, seed=13)
    rep, st, dump = orc.ref_replay("log_server", req, dump=True)
    save("log_server", req=raw(req), rep=raw(rep), dump=np.frombuffer(dump, np.uint8), meta=json.dumps({"ring": 1_000_000}))


def gen_store():
    req = tracegen.store_random(12000, seed=14, n_sub_touch=50)
    rep, st = orc.ref_replay("store", req)
    save("store", req=raw(req), rep=raw(rep), meta=json.dumps({"n_sub": STORE_SUB, "hash_size": STORE_SUB * 18 // 4, "touch": 50}))


def gen_tatp():
    touch = 40
    o = orc.TatpOracle(TATP_SUB, populate_n=touch)  # only used to learn which rows exist initially
    existing = [o.dump(t)[0] for t in range(5)]
    req = tracegen.tatp_random(12000, existing, seed=15, n_sub_touch=touch)
    rep, st, dump = orc.ref_replay("tatp", req, dump=True)
    # keep only the lock + log tail of the dump (the table image is 4 GB)
    save("tatp", req=raw(req), rep=raw(rep), meta=json.dumps({"n_sub": TATP_SUB, "touch": touch}),
         dump_tail=np.frombuffer(_tatp_dump_tail(dump), np.uint8))


def _tatp_dump_tail(dump: bytes) -> bytes:
    off = 0
    for _ in range(5):  # skip the 5 kvs dumps: u64 n + n * (8+4+40)
        n = int(np.frombuffer(dump, "<u8", 1, off)[0])
        off += 8 + n * 52
    return dump[off:]


def gen_smallbank():
    req = tracegen.sb_random(20000, seed=16, n_acct_touch=40)
    rep, st, dump = orc.ref_replay("smallbank", req, dump=True)
    off = 0
    for _ in range(2):
        n = int(np.frombuffer(dump, "<u8", 1, off)[0])
        off += 8 + n * 20
    save("smallbank", req=raw(req), rep=raw(rep), meta=json.dumps({"n_acct": SB_ACCT, "touch": 40}),
         dump_tail=np.frombuffer(dump[off:], np.uint8))


GEN = {"lock_fasst": gen_lock_fasst, "lock_2pl": gen_lock_2pl, "log_server": gen_log_server,
       "store": gen_store, "tatp": gen_tatp, "smallbank": gen_smallbank}

if __name__ == "__main__":
    orc.build()
    for w in sys.argv[1:] or list(GEN):
        t = time.time()
        GEN[w]()
        print(f"{w}: {time.time()-t:.1f}s")
