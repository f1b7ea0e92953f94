#!/usr/bin/env python3
"""Long reference-made traces for the workloads whose committed .npz fixtures are short: lock_2pl,
log_server, store, smallbank -- and tatp (the closed loop of the restated clients at the reference's 7M subscribers).  Each trace -- 3,000,000 requests -- is generated from seeds by code in this repository
(tests/long_traces.py: the restated lock_2pl / smallbank clients in closed loop against CPU oracle servers, seeded
streams for the other two), replayed through the UNMODIFIED reference udp/ server (oracle/_ref/ref_*, compile-time
sizes) and only hashes are committed: of the request stream, of the reference's reply stream (whole and a 1M prefix)
and of its state dump where the harness writes one.  tests/test_long_traces.py regenerates the traces -- through the CPU
oracle, and on the GPU through the engines -- and compares.  Needs /root/reference (`make -C oracle ref`).

    python tests/golden/make_long.py [workload ...]
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import long_traces as lt  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def sha(a) -> str:
    return hashlib.sha256(a.tobytes() if hasattr(a, "tobytes") else a).hexdigest()


def main():
    path = os.path.join(HERE, "long_traces.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for wl in sys.argv[1:] or list(lt.TRACES):
        t = time.time()
        servers = lt.oracle_servers(wl)
        req, rep = lt.TRACES[wl](servers)  # closed loop / stream against the CPU oracle
        ref = orc.ref_replay(lt.REF_NAME[wl], req, dump=True)
        ref_rep = ref[0]
        if wl in ("store", "smallbank", "tatp"):  # rows the trace never wrote still hold the reference's populate-time stack bytes
            a, b = orc.mask_populate_garbage(wl, ref_rep.copy()), orc.mask_populate_garbage(wl, rep.copy())
        else:
            a, b = ref_rep, rep
        assert a.tobytes() == b.tobytes(), f"{wl}: the restatement and the unmodified reference disagree"
        if wl in lt.VS:  # ... and the state the N requests leave behind (a fresh oracle server: the closed loops ran past N)
            del servers
            srv = lt.fresh_oracle(wl)
            assert srv.submit(req).tobytes() == rep.tobytes()
            assert lt.digest_of_oracle(wl, srv.o, lt.n_log_appends(wl, rep)) == lt.digest_of_reference_dump(wl, ref[2]), f"{wl}: final states differ"
            del srv
        out[wl] = {"n_requests": len(req), "req_sha256": sha(req), "rep_sha256": sha(b), "rep_prefix_1m_sha256": sha(b[:1 << 20]),
                   # the state the reference is left in: the raw dump of the lock table / the ring; for the kv servers its
                   # canonical form (long_traces.digest_of_reference_dump: rows in bucket / chain order, lock words, log ring)
                   "dump_sha256": lt.digest_of_reference_dump(wl, ref[2]) if wl in lt.VS else sha(ref[2]),
                   "reference_ops_per_s": ref[1].get("ops_per_s"),
                   "reply_types": lt.reply_types(wl, b), "params": lt.PARAMS[wl]}
        print(wl, f"{time.time() - t:.0f}s", json.dumps(out[wl])[:300])
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
