"""The host boundary of the C ABI: dint_submit_async / dint_wait (pipelined H2D / kernels / D2H), stream ordering of
dint_submit_device across caller streams, and the overflow-pool accounting."""
import numpy as np
import pytest
import torch

import tracegen
from dint_amd import _lib, wire
from dint_amd.engine import Engine, Pinned
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
W = wire.Workload


def test_async_tickets_pipeline_and_match_sync_submit():
    n, chunks = 40_000, 7
    reqs = [tracegen.fasst_random(n, seed=10 + k, n_hot=32, p_hot=0.7) for k in range(chunks)]
    a = Engine(W.FASST, n_slots=1 << 20)
    want = [a.submit(r) for r in reqs]
    b = Engine(W.FASST, n_slots=1 << 20, max_pass=4096)  # ten passes per submission, three staging slots
    pins = [(Pinned(n * 9), Pinned(n * 9)) for _ in range(chunks)]
    for (pi, _), r in zip(pins, reqs):
        pi.array[:] = np.frombuffer(r.tobytes(), np.uint8)
    tickets = [b.submit_async(pi.ptr, n, po.ptr) for pi, po in pins]  # all in flight before the first wait
    assert tickets == sorted(tickets) and len(set(tickets)) == chunks
    for t in reversed(tickets):  # waiting out of order is fine
        b.wait(t)
    for (_, po), w in zip(pins, want):
        assert po.array.tobytes() == w.tobytes()
    assert all((x == y).all() for x, y in zip(a.read_locks(), b.read_locks()))
    with pytest.raises(_lib.DintError):
        b.wait(tickets[-1] + 100)


def test_async_pageable_buffers_and_in_place():
    o = orc.TatpOracle(300, log_entries=20_000)
    req = tracegen.tatp_random(30_000, [o.dump(t)[0] for t in range(5)], seed=2, n_sub_touch=50)
    e = Engine(W.TATP, n_rows=300, log_entries=20_000, max_pass=8192)
    e.populate(300)
    buf = req.copy()
    t = e.submit_async(buf, len(buf), buf)  # pageable numpy memory, replies over the requests
    e.wait(t)
    assert buf.tobytes() == o.replay(req).tobytes()


def test_pageable_callers_three_engines_back_to_back():
    """The host boundary with plain (pageable) buffers, as the reference's stack `message` (lock_fasst/udp/net.h:33-48): three
    engines in one process, many batches back to back of sizes that end inside a 4 KB page and at odd alignments, several
    passes per batch, mixed with page-locked submissions.  Every reply byte must equal the oracle's -- r04 saw the tail of a
    page of one reply batch hold another batch's bytes when the HIP runtime copied pageable memory under a profiler; the
    engine stages pageable callers through its own page-locked buffers now."""
    rng = np.random.default_rng(41)
    n_eng, rounds = 3, 40
    orcs = [orc.SmallbankOracle(20_000, log_entries=50_000) for _ in range(n_eng)]
    engs = [Engine(W.SMALLBANK, n_rows=20_000, log_entries=50_000, max_pass=4096) for _ in range(n_eng)]
    for e in engs:
        e.populate(20_000)
    msg = wire.SB_MSG.itemsize
    pend = []
    for r in range(rounds):
        for k, e in enumerate(engs):
            n = int(rng.integers(150, 9000))
            req = tracegen.sb_random(n, seed=1000 * r + k, n_acct_touch=200)
            want = orcs[k].replay(req)
            # a reply buffer at an odd offset inside a larger pageable allocation, pre-filled with a pattern
            off = int(rng.integers(0, 4096))
            raw = np.full(n * msg + 8192, 0xA5, np.uint8)
            out = raw[off:off + n * msg]
            if r % 5 == 4:  # every fifth round through page-locked memory: both kinds share the staging slots
                pi, po = Pinned(n * msg), Pinned(n * msg)
                pi.array[:] = np.frombuffer(req.tobytes(), np.uint8)
                t = e.submit_async(pi.ptr, n, po.ptr)
                pend.append((e, t, po.array, want, raw, None, (pi, po)))
            else:
                t = e.submit_async(req, n, out.ctypes.data)
                pend.append((e, t, out, want, raw, (off, n * msg), req))
        if r % 4 == 3 or r == rounds - 1:  # several batches per engine in flight before the first wait
            for e, t, out, want, raw, span, _keep in pend:
                e.wait(t)
                assert out.tobytes() == want.tobytes()
                if span:  # nothing outside the reply range was touched
                    assert (raw[:span[0]] == 0xA5).all() and (raw[span[0] + span[1]:] == 0xA5).all()
            pend = []


def test_submit_device_on_alternating_caller_streams_is_serial():
    """consecutive passes on different streams share the engine's scratch: the engine must order them itself"""
    n, steps = 60_000, 12
    reqs = [tracegen.fasst_random(n, seed=50 + k, n_hot=4, p_hot=0.9) for k in range(steps)]
    o = orc.FasstOracle(4801)
    e = Engine(W.FASST, n_slots=4801)
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    d = [torch.from_numpy(np.frombuffer(r.tobytes(), np.uint8).copy()).cuda() for r in reqs]
    torch.cuda.synchronize()
    for k in range(steps):
        e.submit_device(d[k], n, d[k], s[k & 1].cuda_stream if k % 3 else 0)  # also the engine's own stream
    torch.cuda.synchronize()
    e.sync()
    for k in range(steps):
        assert d[k].cpu().numpy().tobytes() == o.replay(reqs[k]).tobytes(), k


@pytest.mark.parametrize("flags", [1, 0])  # 1 = DINT_FLAG_KV_ROUNDS; 0 = the default path (closed forms)
def test_pool_exhaustion_is_reported_not_silent(flags):
    """a full overflow pool refuses INSERTs: every INSERT that stores nothing carries the reject code -- also on the
    default path, where a bucket run that inserts goes request by request once the pool runs low --, DINT_ENOMEM from
    the host submit, nothing stored"""
    S = wire.Store
    e = Engine(W.STORE, n_rows=8, pool_entries=2, flags=flags)  # 36 buckets, 2 overflow entries
    n = 2000
    m = np.zeros(n, wire.STORE_MSG)
    m["type"], m["key"] = S.INSERT, np.arange(1, n + 1, dtype=np.uint64) << np.uint64(32)
    rep = np.empty_like(m)
    with pytest.raises(_lib.DintError, match="pool"):
        _lib.check(e._L.dint_submit(e._h, m.ctypes.data, n, rep.ctypes.data))
    st = e.stats()
    acked, refused = int((rep["type"] == S.INSERT_ACK).sum()), int((rep["type"] == S.REJECT_INSERT).sum())
    assert acked + refused == n and refused == st["pool_exhausted"] > 0
    assert acked == len(e.dump_rows(0)[0]) <= 36 * 4 + 2 * 4
    # a second, harmless batch does not report the old failures again
    rd = m[:10].copy()
    rd["type"] = S.READ
    assert set(e.submit(rd)["type"].tolist()) <= {int(S.GRANT_READ), int(S.NOT_EXIST)}
    # the default pool takes the same inserts without a single refusal
    e2 = Engine(W.STORE, n_rows=8, pool_entries=4096)
    assert (e2.submit(m)["type"] == S.INSERT_ACK).all() and e2.stats()["pool_exhausted"] == 0
