"""The fixed 24M-op lock_fasst trace (BASELINE.json north_star "bit-identical abort/commit outcomes vs the reference on
a fixed 24M-op trace").

tests/golden/fasst_24m.json holds hashes recorded from the UNMODIFIED lock_fasst/udp/server.cc (36M slots) replaying the
24,000,000 requests that the restated reference client (lock_fasst/caladan/client.cc:183-280, 4096 workers, 24M lids,
read proportion 0.8, Zipf-0.8 and uniform keys) issues in closed loop -- see tests/golden/make_fasst_24m.py.  The trace is
regenerated here, never stored: the client's next request depends on every reply (REJECT -> ABORT what was locked ->
restart; changed version -> roll back), so reproducing the request hash proves every grant / reject / version the
server returned on the way, and the counters are the abort / commit outcomes."""
import hashlib
import json
import os

import numpy as np
import pytest

from dint_amd import wire
from dint_amd.driver import fasst_trace
from oracle import oracle as orc

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fasst_24m.json")))
N = FIX["n_requests"]
KW = dict(n_workers=FIX["n_workers"], key_space=FIX["key_space"], read_pct=FIX["read_pct"])


def sha(a) -> str:
    return hashlib.sha256(a.tobytes()).hexdigest()


class OracleServer:
    def __init__(self):
        self.o = orc.FasstOracle(FIX["n_slots"])

    def submit(self, req):
        return self.o.replay(req)


def dump_bytes(lock: np.ndarray, ver: np.ndarray) -> bytes:
    """the reference harness's state dump: u32 count, then {slot, lock, ver} of every non-zero slot"""
    nz = np.nonzero(lock | ver)[0].astype("<u4")
    rows = np.stack([nz, lock[nz].astype("<u4"), ver[nz].astype("<u4")], axis=1)
    return np.array([len(nz)], "<u4").tobytes() + rows.tobytes()


@pytest.mark.parametrize("variant", ["zipf0.8", "uniform"])
def test_oracle_closed_loop_reproduces_the_trace_prefix(variant):
    v = FIX["variants"][variant]
    p = 2_097_152
    req, rep, st = fasst_trace(OracleServer(), p, zipf_theta=v["zipf_theta"], **KW)
    assert st["protocol_errors"] == 0
    assert sha(req) == v["req_prefix_sha256"][str(p)] and sha(rep) == v["rep_prefix_sha256"][str(p)]
    assert sha(req[:262_144]) == v["req_prefix_sha256"]["262144"]


def test_oracle_closed_loop_full_24m_matches_the_unmodified_reference():
    v = FIX["variants"]["zipf0.8"]
    srv = OracleServer()
    req, rep, st = fasst_trace(srv, N, zipf_theta=0.8, **KW)
    assert sha(req) == v["req_sha256"] and sha(rep) == v["rep_sha256"]
    assert {k: st[k] for k in v["client"]} == v["client"] and st["protocol_errors"] == 0
    assert hashlib.sha256(dump_bytes(srv.o.locks, srv.o.vers)).hexdigest() == v["dump_sha256"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("variant", ["zipf0.8", "uniform"])
def test_gpu_24m_trace_bit_identical_to_the_reference(variant):
    from dint_amd.engine import Engine

    v = FIX["variants"][variant]
    eng = Engine(wire.Workload.FASST, n_slots=FIX["n_slots"])
    # closed loop, one 4096-request batch per epoch: requests AND replies must hash to the reference's
    req, rep, st = fasst_trace(eng, N, zipf_theta=v["zipf_theta"], **KW)
    assert st["protocol_errors"] == 0
    assert sha(req) == v["req_sha256"], "a grant / reject / version differed somewhere: the clients took another path"
    assert sha(rep) == v["rep_sha256"]
    assert {k: st[k] for k in v["client"]} == v["client"]  # abort / commit outcomes
    types = np.bincount(rep["type"], minlength=9)
    assert {str(k): int(c) for k, c in enumerate(types) if c} == v["reply_types"]
    lock, ver = eng.read_locks()
    assert hashlib.sha256(dump_bytes(lock, ver)).hexdigest() == v["dump_sha256"]
    del eng
    # the fixed trace replayed open loop at other batch sizes: 65,536 (the whole trace), 1M (kernel passes split
    # inside the engine) and 64 (the first 2M requests)
    big = Engine(wire.Workload.FASST, n_slots=FIX["n_slots"])
    got = np.concatenate([big.submit(req[i:i + 65_536]) for i in range(0, N, 65_536)])
    assert sha(got) == v["rep_sha256"]
    lock, ver = big.read_locks()
    assert hashlib.sha256(dump_bytes(lock, ver)).hexdigest() == v["dump_sha256"]
    big.reset()
    got = np.concatenate([big.submit(req[i:i + (1 << 20)]) for i in range(0, 8_388_608, 1 << 20)])
    assert sha(got) == v["rep_prefix_sha256"]["8388608"]
    big.reset()
    p = 2_097_152
    got = np.concatenate([big.submit(req[i:i + 64]) for i in range(0, p, 64)])
    assert sha(got) == v["rep_prefix_sha256"][str(p)]
