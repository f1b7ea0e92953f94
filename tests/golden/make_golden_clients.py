#!/usr/bin/env python3
"""Generate tests/golden/clients.npz by running the UNMODIFIED reference clients -- <wl>/caladan/client_udp_shard.cc, the
seven TATP and six SmallBank transactions -- one client at a time against three CPU oracle shard
servers (oracle/ref_harness/caladan: `make -C oracle ref_client` compiles the client translation unit against a
synchronous stand-in for the Caladan runtime, whose submodules are empty in the reference tree).  Only runs where
/root/reference exists; the fixture is committed.

    python tests/golden/make_golden_clients.py

Per client (workload, worker gid) and shard server the fixture holds the requests the client sent and the replies it
got, in order.  Every 7th lock request was refused by the harness, so the abort paths are in the streams.  Bytes the
reference leaves unassigned (its `message` structs are uninitialised stack objects: `val` / `ver` of requests without a
payload, `ord` of single-message phases) are zeroed on both sides -- dint_amd/csrc/txn_clients.h sends zeros there.
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

CLIENTS = {"tatp": (0, 5, 123456), "smallbank": (0, 9, 777)}
MESSAGES = 12000
PAYLOAD_REQ = {"tatp": (12, 13, 14, 18, 19), "smallbank": (4, 5, 6)}   # requests that carry val + ver
PAYLOAD_REP = {"tatp": (4,), "smallbank": (7, 9)}                      # replies whose val + ver the server assigns


def canon(wl, req, rep):
    req, rep = req.copy(), rep.copy()
    keep = np.isin(req["type"], PAYLOAD_REQ[wl])
    req["val"][~keep], req["ver"][~keep], req["ord"] = 0, 0, 0
    if wl == "tatp":  # a new call-forwarding row: only end_time and numberx[0] are assigned (client_udp_shard.cc:843-845)
        v = req["val"].copy()
        v[keep & (req["table"] == 4), 2:] = 0
        req["val"] = v
    keep_r = np.isin(rep["type"], PAYLOAD_REP[wl]) | keep
    rep["val"][~keep_r], rep["ver"][~keep_r], rep["ord"] = 0, 0, 0
    return req, rep


if __name__ == "__main__":
    out, meta = {}, {}
    with tempfile.TemporaryDirectory(prefix="dint_clients_") as td:
        for wl, gids in CLIENTS.items():
            dt = wire.TATP_MSG if wl == "tatp" else wire.SB_MSG
            for g in gids:
                pre = os.path.join(td, f"{wl}_{g}")
                r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", f"ref_client_{wl}"), str(g), str(MESSAGES), pre],
                                   capture_output=True, text=True, check=True)
                meta[f"{wl}_{g}"] = json.loads(r.stdout.strip().splitlines()[-1])
                for s in range(3):
                    req, rep = canon(wl, np.fromfile(f"{pre}.s{s}.req", dt), np.fromfile(f"{pre}.s{s}.rep", dt))
                    out[f"{wl}_{g}_s{s}_req"] = np.frombuffer(req.tobytes(), np.uint8)
                    out[f"{wl}_{g}_s{s}_rep"] = np.frombuffer(rep.tobytes(), np.uint8)
                print(wl, g, meta[f"{wl}_{g}"])
    path = os.path.join(HERE, "clients.npz")
    np.savez_compressed(path, meta=json.dumps({"clients": {k: list(v) for k, v in CLIENTS.items()}, "runs": meta,
                                               "n_rows": {"tatp": 7_000_000, "smallbank": 24_000_000}}), **out)
    print(f"wrote {path} ({os.path.getsize(path)/1e6:.2f} MB)")
