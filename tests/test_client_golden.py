"""The transaction handlers against the reference's own clients.

tests/golden/clients.npz holds what the UNMODIFIED tatp / smallbank `caladan/client_udp_shard.cc` sent to, and got from,
three shard servers -- one client at a time, 12,000 messages each, with every 7th lock request refused so that the abort
paths run (tests/golden/make_golden_clients.py; the client translation units compile against a stand-in for the Caladan
runtime, oracle/ref_harness/caladan).  The restated client state machines of dint_amd/csrc/txn_clients.h, fed the same
replies, must send the same requests: same transaction draw (seed 0xdeadbeef + gid, mix 35/35/10/2/14/2/2 resp.
15/15/15/25/15/15), same keys, same messages to the same shard in the same order, same values and versions on every
request that carries them, same reaction to NOT_EXIST / REJECT.  (The GPU-resident clients of k_txn.hip are the same
source compiled for the device and are held bit-identical to this host driver by tests/test_gpu_gdriver.py.)"""
import json
import os

import numpy as np
import pytest

from dint_amd import wire
from dint_amd.driver import Driver

G = os.path.join(os.path.dirname(__file__), "golden")
PAYLOAD_REQ = {"tatp": (12, 13, 14, 18, 19), "smallbank": (4, 5, 6)}  # requests that carry val + ver


def _canon(wl, a):
    """the bytes the reference assigns (its request structs are uninitialised stack objects otherwise)"""
    a = a.copy()
    keep = np.isin(a["type"], PAYLOAD_REQ[wl])
    a["val"][~keep], a["ver"][~keep], a["ord"] = 0, 0, 0
    if wl == "tatp":  # INSERT_CALL_FORWARDING builds its row in the buffer of the READ that came back NOT_EXIST and assigns
        cf = keep & (a["table"] == 4)  # end_time and numberx[0] only (client_udp_shard.cc:843-845): the rest is whatever
        v = a["val"].copy()            # the uninitialised READ request held
        v[cf, 2:] = 0
        a["val"] = v
    return a


@pytest.mark.parametrize("wl", ["tatp", "smallbank"])
def test_restated_clients_send_what_the_reference_clients_send(wl):
    z = np.load(os.path.join(G, "clients.npz"))
    meta = json.loads(str(z["meta"]))
    W = wire.Workload.TATP if wl == "tatp" else wire.Workload.SMALLBANK
    dt = wire.MSG_DTYPE[W]
    for gid in meta["clients"][wl]:
        req = [np.frombuffer(z[f"{wl}_{gid}_s{s}_req"].tobytes(), dt) for s in range(3)]
        rep = [np.frombuffer(z[f"{wl}_{gid}_s{s}_rep"].tobytes(), dt) for s in range(3)]
        d = Driver(W, 1, meta["n_rows"][wl], first_client=gid)  # one client; key_dist = the reference's own
        cur, epochs = [0, 0, 0], 0
        while True:
            out = d.next()
            if any(cur[s] + len(out[s]) > len(req[s]) for s in range(3)):
                break  # the recording was cut inside this phase
            for s in range(3):
                want = _canon(wl, req[s][cur[s]:cur[s] + len(out[s])])
                assert _canon(wl, out[s]).tobytes() == want.tobytes(), (wl, gid, epochs, s)
            d.consume([rep[s][cur[s]:cur[s] + len(out[s])].copy() for s in range(3)])
            for s in range(3):
                cur[s] += len(out[s])
            epochs += 1
        st = d.stats()
        nt = 7 if wl == "tatp" else 6
        assert sum(cur) > 0.99 * meta["runs"][f"{wl}_{gid}"]["messages"]
        assert all(c > 0 for c in st["by_type"][:nt]) and st["committed"] < st["txns"]  # every transaction type, aborts too
        assert meta["runs"][f"{wl}_{gid}"]["locks_refused"] > 100


def test_restated_micro_clients_send_what_the_reference_load_generators_send():
    """tests/golden/clients_micro.npz (make_golden_clients_micro.py): one worker each of the UNMODIFIED
    lock_fasst/caladan/client.cc and lock_2pl/caladan/client.cc, replaying the transactions the restated clients draw for
    their worker 0, against a CPU oracle lock server, with one ACQUIRE in five refused by the harness (aborts, releases,
    restarts, validation).  dint_amd/csrc/fasst_client.cc and dint_amd/driver.py::TplClient, fed the recorded replies one by
    one, must send the recorded requests -- the "fixed 24M-op trace" of tests/test_fasst_24m.py is made by this client."""
    from dint_amd.driver import FasstClient, TplClient

    z = np.load(os.path.join(G, "clients_micro.npz"))
    meta = json.loads(str(z["meta"]))
    req = np.frombuffer(z["fasst_req"].tobytes(), wire.FASST_MSG)
    rep = np.frombuffer(z["fasst_rep"].tobytes(), wire.FASST_MSG)
    c = FasstClient(1, meta["fasst"]["key_space"], zipf_theta=None)
    for i in range(len(req)):
        out = c.next()
        # (the reference's requests carry ver = 0: `message msg = {type, lid, 0}`, client.cc:199,216,225,237,248,258)
        assert out.tobytes() == req[i:i + 1].tobytes(), ("lock_fasst", i, out, req[i])
        c.consume(rep[i:i + 1].copy())
    st = c.stats()
    assert st["protocol_errors"] == 0 and st["rejects"] > 300 and st["committed"] > 500 and st["rollbacks"] == 0
    assert len(req) == meta["fasst"]["messages"] and set(rep["type"].tolist()) == {4, 5, 6, 7, 8}

    req = np.frombuffer(z["tpl_req"].tobytes(), wire.TPL_MSG)
    rep = np.frombuffer(z["tpl_rep"].tobytes(), wire.TPL_MSG)
    t = TplClient(1, meta["tpl"]["key_space"], zipf_theta=None, seed=meta["tpl"]["seed"])
    for i in range(len(req)):
        out = t.next()
        assert out.tobytes() == req[i:i + 1].tobytes(), ("lock_2pl", i, out, req[i])
        t.consume(rep[i:i + 1].copy())
    st = t.stats()
    assert st["protocol_errors"] == 0 and st["rejects"] > 1000 and st["committed"] > 500
    assert len(req) == meta["tpl"]["messages"] and set(rep["action"].tolist()) == {2, 3, 5}
