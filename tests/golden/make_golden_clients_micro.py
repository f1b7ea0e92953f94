#!/usr/bin/env python3
"""Generate tests/golden/clients_micro.npz: the request / reply streams of the UNMODIFIED reference load generators of the
two lock micro-benchmarks -- lock_fasst/caladan/client.cc and lock_2pl/caladan/client.cc, one worker each -- run against a
CPU oracle lock server by oracle/ref_harness/caladan/ref_client_micro.cc (`make -C oracle ref_client`: the client
translation unit compiled against a synchronous stand-in for the Caladan runtime).  VERDICT r03 item 7b: the pin of
dint_amd/csrc/fasst_client.cc and dint_amd/driver.py::TplClient.  Only runs where /root/reference exists; the fixture is
committed.

    python tests/golden/make_golden_clients_micro.py

The reference clients read their transactions from a trace file (its trace_init.sh writes them from unseeded Python
random numbers, so none ships with it).  Here the file holds the transactions the RESTATED client draws for its worker 0
(peeked one by one while it runs against a server that grants everything); the reference client then replays them with
every 5th ACQUIRE refused by the harness -- aborts, releases, restarts -- and what it sent and received is recorded.
tests/test_client_golden.py feeds the recorded replies to a fresh restated client and demands the recorded requests.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from dint_amd import wire  # noqa: E402
from dint_amd.driver import FasstClient, TplClient  # noqa: E402

N_TXN, MESSAGES, REFUSE_EVERY = 1500, 30000, 5
KEY_SPACE = 24_000_000
TPL_SEED = 0xDEADBEEF


def fasst_transactions():
    """worker 0's first N_TXN transactions: run the restated client against a server that grants and never changes a
    version; every commit draws the next transaction"""
    c = FasstClient(1, KEY_SPACE, zipf_theta=None)
    txns, done = [c.peek(0)], 0
    while len(txns) < N_TXN:
        m = c.next()
        m["type"] = {0: 4, 1: 5, 2: 7, 3: 8}[int(m["type"][0])]
        c.consume(m)
        if c.stats()["committed"] != done:
            done = c.stats()["committed"]
            txns.append(c.peek(0))
    return txns


def tpl_transactions():
    c = TplClient(1, KEY_SPACE, zipf_theta=None, seed=TPL_SEED)
    txns, done = [], -1
    while len(txns) < N_TXN:
        if c.stats()["committed"] != done:
            done = c.stats()["committed"]
            n = int(c.nlock[0])
            txns.append((c.lid[0, :n].tolist(), c.typ[0, :n].tolist()))
        m = c.next()
        m["action"] = np.where(m["action"] == 0, 2, 5)
        c.consume(m)
    return txns


def run(binary, csv_lines, dt):
    with tempfile.TemporaryDirectory(prefix="dint_micro_") as td:
        d = os.path.join(td, "traces", "microbenchmarks", "lock_24000000_r_0.8")
        os.makedirs(d)
        with open(os.path.join(d, "trace_0.csv"), "w") as f:
            f.write("\n".join(csv_lines) + "\n")
        pre = os.path.join(td, "out")
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", binary), str(MESSAGES), pre, str(REFUSE_EVERY)], cwd=td,
                           capture_output=True, text=True, check=True)
        return np.fromfile(pre + ".req", dt), np.fromfile(pre + ".rep", dt), json.loads(r.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    out, meta = {}, {}
    ft = fasst_transactions()
    lines = ["tid,type,lid"]  # lock_fasst/caladan/trace_init.sh:9,24-27: the read set, then the write set
    for tid, (keys, wkeys) in enumerate(ft):
        lines += [f"{tid},0,{k}" for k in keys] + [f"{tid},1,{k}" for k in wkeys]
    req, rep, st = run("ref_client_fasst", lines, wire.FASST_MSG)
    out["fasst_req"], out["fasst_rep"] = np.frombuffer(req.tobytes(), np.uint8), np.frombuffer(rep.tobytes(), np.uint8)
    meta["fasst"] = dict(st, transactions=len(ft), refuse_every=REFUSE_EVERY, key_space=KEY_SPACE,
                         reply_types={str(k): int(v) for k, v in enumerate(np.bincount(rep["type"], minlength=9)) if v})
    tt = tpl_transactions()
    lines = ["txn_id,action,lock_id,lock_type"]  # lock_2pl/caladan/trace_init.sh:9,20-23: acquire in order, release in reverse
    for tid, (lids, types) in enumerate(tt):
        lines += [f"{tid},0,{l},{t}" for l, t in zip(lids, types)] + [f"{tid},1,{l},{t}" for l, t in reversed(list(zip(lids, types)))]
    req, rep, st = run("ref_client_2pl", lines, wire.TPL_MSG)
    out["tpl_req"], out["tpl_rep"] = np.frombuffer(req.tobytes(), np.uint8), np.frombuffer(rep.tobytes(), np.uint8)
    meta["tpl"] = dict(st, transactions=len(tt), refuse_every=REFUSE_EVERY, key_space=KEY_SPACE, seed=TPL_SEED,
                       reply_actions={str(k): int(v) for k, v in enumerate(np.bincount(rep["action"], minlength=6)) if v})
    print(json.dumps(meta, indent=1))
    path = os.path.join(HERE, "clients_micro.npz")
    np.savez_compressed(path, meta=json.dumps(meta), **out)
    print(f"wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")
