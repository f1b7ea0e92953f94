#!/usr/bin/env python3
"""The fixed 24M-op lock_fasst trace (BASELINE.json north_star) and its reference answers.

The trace is generated, not stored (216 MB): dint_amd.driver.fasst_trace = the reference ClientLoop
(lock_fasst/caladan/client.cc:183-280) restated with 4096 workers over 24,000,000 lids, read proportion 0.8,
keys ~ Zipf(0.8) (and the reference's uniform variant), driven here by the CPU oracle.  The 24,000,000 requests are then
replayed through the UNMODIFIED reference server (oracle/_ref/ref_lock_fasst, 36,000,000 slots) and this script
commits only hashes: of the request stream (whole and prefixes), of the reference's reply stream, of its final
{slot, lock, ver} dump, plus the abort / commit outcome counters.  tests/test_fasst_24m.py regenerates the trace
through the GPU engine and compares.  Needs /root/reference (run `make -C oracle ref` first).

    python tests/golden/make_fasst_24m.py
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from dint_amd.driver import fasst_trace  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N = 24_000_000
PREFIXES = [262_144, 2_097_152, 8_388_608]


class OracleServer:
    def __init__(self, nslots):
        self.o = orc.FasstOracle(nslots)

    def submit(self, req):
        return self.o.replay(req)


def sha(a) -> str:
    return hashlib.sha256(a.tobytes() if hasattr(a, "tobytes") else a).hexdigest()


def main():
    out = {"n_requests": N, "n_workers": 4096, "key_space": 24_000_000, "read_pct": 80, "n_slots": 36_000_000, "variants": {}}
    for name, theta in (("zipf0.8", 0.8), ("uniform", None)):
        t = time.time()
        req, rep, st = fasst_trace(OracleServer(36_000_000), N, n_workers=4096, key_space=24_000_000, zipf_theta=theta)
        t_gen = time.time() - t
        assert st["protocol_errors"] == 0
        ref_rep, ref_st, dump = orc.ref_replay("lock_fasst", req, dump=True)
        assert ref_rep.tobytes() == rep.tobytes(), "the restatement and the unmodified reference disagree"
        types = np.bincount(rep["type"], minlength=9)
        out["variants"][name] = {
            "zipf_theta": theta, "req_sha256": sha(req), "rep_sha256": sha(ref_rep), "dump_sha256": sha(dump),
            "dump_nonzero_slots": int(np.frombuffer(dump[:4], "<u4")[0]),
            "req_prefix_sha256": {str(p): sha(req[:p]) for p in PREFIXES},
            "rep_prefix_sha256": {str(p): sha(ref_rep[:p]) for p in PREFIXES},
            "reply_types": {str(k): int(v) for k, v in enumerate(types) if v},
            "client": {k: st[k] for k in ("committed", "rejects", "rollbacks", "epochs")},
            "reference_ops_per_s": ref_st.get("ops_per_s"), "gen_s": round(t_gen, 1),
        }
        print(name, json.dumps(out["variants"][name])[:400])
    with open(os.path.join(HERE, "fasst_24m.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
