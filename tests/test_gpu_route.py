"""GPU tests of the multi-GPU path on ONE GPU: the HIP pack / unpack kernels against their numpy
restatement, segmented passes against contiguous ones, the whole Router with a self all-to-all, and two REAL
sharded engine groups (two ranks sharing the GPU, exchange staged through the host over gloo) against the
unsharded CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

import shard_double as sd
import tracegen
from dint_amd import wire
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = wire.Workload


def _engine(*a, **k):
    from dint_amd.engine import Engine

    return Engine(*a, **k)


def _dev(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.frombuffer(a.tobytes(), np.uint8).copy()).cuda()


# ------------------------------------------------------------------------------------------- pack / unpack
@pytest.mark.parametrize("n,world,cap", [(1, 2, 64), (5000, 2, 4096), (70_000, 8, 9000), (70_000, 8, 8000), (1 << 20, 3, 400_000),
                                         (0, 4, 64), (1025, 64, 64)])
def test_route_pack_unpack_vs_numpy(n, world, cap):
    nslots, rank = 36_000_000, 1
    req = tracegen.fasst_random(max(n, 1), seed=n + world, n_hot=8, p_hot=0.3)[:n]
    eng = _engine(W.FASST, n_slots=nslots, shard_index=rank, shard_count=world)
    msg, HDRB = 9, 64
    stride = HDRB + (cap * msg + 15) // 16 * 16
    d_req = _dev(req) if n else torch.zeros(16, dtype=torch.uint8, device="cuda")
    d_send = torch.zeros(world * stride, dtype=torch.uint8, device="cuda")
    d_slot = torch.full((max(n, 1),), 12345, dtype=torch.int32, device="cuda")
    eng.route_pack(d_req, n, d_send.data_ptr() + HDRB, cap, stride, d_send.data_ptr(), stride, d_slot)
    eng.sync()
    # the numpy restatement on host buffers
    dbl = sd.ServerDouble(W.FASST, None, world, rank, sd.lid_home(nslots, world))
    h_req = torch.from_numpy(np.frombuffer(req.tobytes(), np.uint8).copy()) if n else torch.zeros(16, dtype=torch.uint8)
    h_send = torch.zeros(world * stride, dtype=torch.uint8)
    h_slot = torch.full((max(n, 1),), 12345, dtype=torch.int32)
    dbl.route_pack(h_req, n, h_send.data_ptr() + HDRB, cap, stride, h_send.data_ptr(), stride, h_slot)
    got = d_send.cpu().numpy().reshape(world, stride)
    want = h_send.numpy().reshape(world, stride)
    cnt = got[:, :4].copy().view("<u4")[:, 0]
    assert (cnt == want[:, :4].copy().view("<u4")[:, 0]).all()
    for w in range(world):
        assert (got[w, HDRB:HDRB + cnt[w] * msg] == want[w, HDRB:HDRB + cnt[w] * msg]).all(), w
    assert (d_slot.cpu().numpy()[:n] == h_slot.numpy()[:n]).all()
    assert eng.stats()["route_overflow"] == dbl.route_overflow
    # unpack straight from the send buffer: every request that was sent comes back unchanged; one that found its slot
    # full comes back with the back-pressure reply (dint_refuse: lock_fasst refuses ACQUIRE_LOCK only)
    d_rep = torch.zeros_like(d_req)
    eng.route_unpack(d_send.data_ptr() + HDRB, cap, stride, d_slot, d_req, n, d_rep)
    eng.sync()
    want_rep = req.copy()
    over = h_slot.numpy()[:n].view(np.uint32) == 0xFFFFFFFF
    if n:
        from dint_amd.engine import refuse

        want_rep[over] = refuse(W.FASST, req[over])
        assert over.sum() == dbl.route_overflow
    assert d_rep.cpu().numpy()[:n * msg].tobytes() == want_rep.tobytes()


@pytest.mark.parametrize("sizes,world", [((5000, 0, 777), 2), ((70_000, 65_000, 1), 8), ((300, 300, 300), 3)])
def test_route_multi_equals_per_engine_calls(sizes, world):
    """dint_route_pack_multi / _unpack_multi (the S servers of a rank in one set of launches, grid.y = server) against
    the same batches routed one engine at a time: identical exchange buffers, slot maps, overflow counts, replies."""
    from dint_amd.engine import route_pack_multi, route_unpack_multi

    rank, msg, HDRB = 1, 55, 64
    o = orc.TatpOracle(300, log_entries=1000)
    existing = [o.dump(t)[0] for t in range(5)]
    reqs = [tracegen.tatp_random(max(n, 1), existing, seed=11 + k, n_sub_touch=200)[:n] for k, n in enumerate(sizes)]
    caps = [max(64, (3 * n) // (2 * world) // 64 * 64 + 64) for n in sizes]
    caps[1] = max(64, sizes[1] // world // 2 // 64 * 64)  # the second server's slots overflow
    offs, o_ = [], HDRB
    for c in caps:
        offs.append(o_)
        o_ += (c * msg + 15) // 16 * 16
    stride = (o_ + 63) // 64 * 64

    def run(multi):
        engs = [_engine(W.TATP, n_rows=300 * world, log_entries=1000, shard_index=rank, shard_count=world) for _ in sizes]
        d_req = [_dev(r) if len(r) else torch.zeros(16, dtype=torch.uint8, device="cuda") for r in reqs]
        d_send = torch.zeros(world * stride, dtype=torch.uint8, device="cuda")
        d_slot = [torch.full((max(n, 1),), 12345, dtype=torch.int32, device="cuda") for n in sizes]
        d_rep = [torch.zeros_like(d) for d in d_req]
        sp = d_send.data_ptr()
        if multi:
            route_pack_multi(engs, d_req, list(sizes), [sp + offs[k] for k in range(3)], caps, stride,
                             [sp + 4 * k for k in range(3)], stride, d_slot)
            route_unpack_multi(engs, [sp + offs[k] for k in range(3)], caps, stride, d_slot, d_req, list(sizes), d_rep)
        else:
            for k, e in enumerate(engs):
                e.route_pack(d_req[k], sizes[k], sp + offs[k], caps[k], stride, sp + 4 * k, stride, d_slot[k])
            for k, e in enumerate(engs):
                e.route_unpack(sp + offs[k], caps[k], stride, d_slot[k], d_req[k], sizes[k], d_rep[k])
        torch.cuda.synchronize()
        for e in engs:
            e.sync()
        return (d_send.cpu().numpy(), [d.cpu().numpy()[:n] for d, n in zip(d_slot, sizes)],
                [d.cpu().numpy()[:n * msg].tobytes() for d, n in zip(d_rep, sizes)], [e.stats()["route_overflow"] for e in engs])

    a, b = run(False), run(True)
    sa, sb = a[0].reshape(world, stride), b[0].reshape(world, stride)
    cnt = sa[:, :12].copy().view("<u4")
    assert (cnt == sb[:, :12].copy().view("<u4")).all()
    for w in range(world):
        for k in range(3):
            lo = offs[k]
            assert (sa[w, lo:lo + cnt[w, k] * msg] == sb[w, lo:lo + cnt[w, k] * msg]).all(), (w, k)
    assert all((x == y).all() for x, y in zip(a[1], b[1]))
    # straight back from the send buffer: reply = request, except what found its slot full: the back-pressure reply
    from dint_amd.engine import refuse

    want = []
    for k, r in enumerate(reqs):
        w = r.copy()
        over = a[1][k].view(np.uint32) == 0xFFFFFFFF
        w[over] = refuse(W.TATP, r[over])
        want.append(w.tobytes())
    assert a[2] == b[2] and a[2] == want
    assert a[3] == b[3] and a[3][0] == 0 and (a[3][1] > 0) == (sizes[1] > 0)  # only the second server's slots are too small


def _segmented(req: np.ndarray, cuts, cap, hdr=64):
    """lay `req` out as len(cuts)-1 segments of capacity `cap` with a header in front of each"""
    msg = req.dtype.itemsize
    stride = hdr + (cap * msg + 63) // 64 * 64
    nseg = len(cuts) - 1
    buf = np.full(nseg * stride, 0xCD, np.uint8)  # padding slots hold garbage
    for k in range(nseg):
        part = req[cuts[k]:cuts[k + 1]]
        buf[k * stride:k * stride + 4] = np.array([len(part)], "<u4").view(np.uint8)
        buf[k * stride + hdr:k * stride + hdr + len(part) * msg] = np.frombuffer(part.tobytes(), np.uint8)
    return buf, stride, nseg


def _unsegment(buf, cuts, stride, dtype, hdr=64):
    msg = dtype.itemsize
    return np.concatenate([np.frombuffer(buf[k * stride + hdr:k * stride + hdr + (cuts[k + 1] - cuts[k]) * msg].tobytes(), dtype)
                           for k in range(len(cuts) - 1)])


@pytest.mark.parametrize("wl", ["fasst", "tpl", "tatp", "smallbank"])
@pytest.mark.parametrize("cuts,cap", [([0, 3000, 3000, 9000, 20_000], 12_000), ([0, 1, 2, 20_000], 65_536)])
def test_submit_segments_equals_contiguous(wl, cuts, cap):
    n = cuts[-1]
    if wl == "fasst":
        req, mk = tracegen.fasst_random(n, seed=3, n_hot=8, p_hot=0.6), lambda **k: _engine(W.FASST, n_slots=4801, **k)
    elif wl == "tpl":
        req, mk = tracegen.tpl_random(n, seed=4, n_hot=8, p_hot=0.6), lambda **k: _engine(W.TPL, n_slots=4801, **k)
    elif wl == "tatp":
        o = orc.TatpOracle(300, log_entries=200_000)
        req = tracegen.tatp_random(n, [o.dump(t)[0] for t in range(5)], seed=5, n_sub_touch=40)

        def mk(**k):
            e = _engine(W.TATP, n_rows=300, log_entries=200_000, **k)
            e.populate(300)
            return e
    else:
        req = tracegen.sb_random(n, seed=6, n_acct_touch=30)

        def mk(**k):
            e = _engine(W.SMALLBANK, n_rows=2000, log_entries=200_000, **k)
            e.populate(2000)
            return e
    a, b = mk(), mk(max_pass=16_384 if cap <= 16_384 else 0)
    want = a.submit(req)
    buf, stride, nseg = _segmented(req, cuts, cap)
    d = torch.from_numpy(buf).cuda()
    b.submit_segments(d.data_ptr() + 64, nseg, cap, stride, d.data_ptr(), stride)
    b.sync()
    out = d.cpu().numpy()
    got = _unsegment(out, cuts, stride, req.dtype)
    assert got.tobytes() == want.tobytes()
    # padding slots were never written
    for k in range(nseg):
        used = 64 + (cuts[k + 1] - cuts[k]) * req.dtype.itemsize
        assert (out[k * stride + used:(k + 1) * stride] == 0xCD).all()
    if wl in ("tatp", "smallbank"):
        for t in range(5 if wl == "tatp" else 2):
            assert all((x == y).all() for x, y in zip(a.dump_rows(t), b.dump_rows(t)))
        ra, ta = a.read_log(200_000)
        rb, tb = b.read_log(200_000)
        assert ta == tb and ra.tobytes() == rb.tobytes()
        assert b.stats()["bad_requests"] == a.stats()["bad_requests"]
    else:
        assert all((x == y).all() for x, y in zip(a.read_locks(), b.read_locks()))


@pytest.mark.parametrize("wl", ["tatp", "smallbank", "fasst"])
def test_submit_segments_multi_equals_one_call_per_engine(wl):
    """dint_submit_segments_multi: three engines' segments in one set of launches (grid.y = engine) on the caller's
    stream -- replies, tables, locks and log rings must equal three dint_submit_segments calls; different batch sizes per
    engine, one empty segment, two rounds (the scratch lists alternate per pass).  Lock engines take the fallback (one
    call per engine on that stream)."""
    from dint_amd.engine import submit_segments_multi

    sizes = [20_000, 7000, 13_000]
    cuts = [[0, n // 3, n // 3, n] for n in sizes]
    cap = 16_384
    if wl == "tatp":
        o = orc.TatpOracle(300, log_entries=200_000)
        rows = [o.dump(t)[0] for t in range(5)]
        gen = lambda k, r: tracegen.tatp_random(sizes[k], rows, seed=50 + 10 * r + k, n_sub_touch=40)  # noqa: E731

        def mk():
            e = _engine(W.TATP, n_rows=300, log_entries=200_000)
            e.populate(300)
            return e
    elif wl == "smallbank":
        gen = lambda k, r: tracegen.sb_random(sizes[k], seed=60 + 10 * r + k, n_acct_touch=30)  # noqa: E731

        def mk():
            e = _engine(W.SMALLBANK, n_rows=2000, log_entries=200_000)
            e.populate(2000)
            return e
    else:
        gen = lambda k, r: tracegen.fasst_random(sizes[k], seed=70 + 10 * r + k, n_hot=8, p_hot=0.6)  # noqa: E731
        mk = lambda: _engine(W.FASST, n_slots=4801)  # noqa: E731
    ones, multi = [mk() for _ in range(3)], [mk() for _ in range(3)]
    st = torch.cuda.Stream()
    for r in range(2):
        reqs = [gen(k, r) for k in range(3)]
        segs = [_segmented(reqs[k], cuts[k], cap) for k in range(3)]
        stride = segs[0][1]
        assert all(sg[1] == stride and sg[2] == 3 for sg in segs)
        da = [torch.from_numpy(sg[0]).cuda() for sg in segs]
        db = [torch.from_numpy(sg[0]).cuda() for sg in segs]
        for k in range(3):
            ones[k].submit_segments(da[k].data_ptr() + 64, 3, cap, stride, da[k].data_ptr(), stride)
            ones[k].sync()
        st.wait_stream(torch.cuda.current_stream())
        submit_segments_multi(multi, [d.data_ptr() + 64 for d in db], 3, [cap] * 3, stride, [d.data_ptr() for d in db], stride,
                              st.cuda_stream)
        st.synchronize()
        for k in range(3):
            assert torch.equal(da[k], db[k]), (r, k)
    for k in range(3):
        if wl == "fasst":
            assert all((x == y).all() for x, y in zip(ones[k].read_locks(), multi[k].read_locks()))
            continue
        for t in range(5 if wl == "tatp" else 2):
            assert all((x == y).all() for x, y in zip(ones[k].dump_rows(t), multi[k].dump_rows(t)))
        ra, ta = ones[k].read_log(200_000)
        rb, tb = multi[k].read_log(200_000)
        assert ta == tb and ra.tobytes() == rb.tobytes()
        sa, sb = ones[k].stats(), multi[k].stats()
        assert sa["bad_requests"] == sb["bad_requests"] and sa["missing_keys"] == sb["missing_keys"]
    # a later per-engine call on the engine's own stream is ordered behind the multi call (order_stream)
    req = gen(0, 5)
    assert ones[0].submit(req).tobytes() == multi[0].submit(req).tobytes()


def test_router_self_exchange_equals_plain_group():
    """world = 1 with the exchange forced on: pack -> self all-to-all -> segments -> unpack must change nothing"""
    from dint_amd.driver import Driver
    from dint_amd.replay import ShardGroup

    n_sub, clients = 20_000, 6000
    plain = ShardGroup(W.TATP, n_sub, log_entries=200_000)
    routed = ShardGroup(W.TATP, n_sub, log_entries=200_000, force_exchange=True, n_max=1 << 16)
    assert routed.router is not None and routed.router.ex.transport == "self"
    d = Driver(W.TATP, clients, n_sub, zipf_theta=0.8)
    for e in range(30):
        req = d.next()
        a, b = plain.submit(req), routed.submit(req)
        for s in range(3):
            assert a[s].tobytes() == b[s].tobytes(), (e, s)
        d.consume(a)
    assert routed.router.overflow() == 0
    for s in range(3):
        for t in range(5):
            assert all((x == y).all() for x, y in zip(plain.engines[s].dump_rows(t), routed.engines[s].dump_rows(t)))


# ------------------------------------------------------------------------- two real ranks on one GPU
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, wl, n_rows, clients, epochs, zipf, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dint_amd.driver import Driver
    from dint_amd.replay import ShardGroup

    torch.cuda.set_device(0)
    grp = ShardGroup(wl, n_rows, device=0, rank=rank, world=world, log_entries=200_000, n_max=1 << 16)
    assert grp.router.ex.transport == "host"
    d = Driver(wl, clients, n_rows, first_client=rank * clients, zipf_theta=zipf)
    trace = []
    for _ in range(epochs):
        req = d.next()
        rep = grp.submit(req)
        d.consume(rep)
        trace.append(([r.tobytes() for r in req], [r.tobytes() for r in rep]))
    # a second pass over the same requests with the tightened slot capacities, device path (what bench.py times)
    caps = grp.router.tighten_caps()
    rows = [[tuple(x.tobytes() for x in e.dump_rows(t)) for t in range(len(e_tables(wl)))] for e in grp.engines]
    st = [e.stats() for e in grp.engines]
    q.put((rank, trace, rows, st, caps, d.stats()))
    dist.barrier()
    dist.destroy_process_group()


def e_tables(wl):
    return range(5) if wire.Workload(wl) == W.TATP else range(2)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("wl,n_rows,clients,zipf", [(int(W.TATP), 20_000, 5000, 0.8), (int(W.SMALLBANK), 50_000, 4000, 0.99 - 1e-9)])
def test_two_sharded_ranks_on_one_gpu_equal_unsharded_oracle(wl, n_rows, clients, zipf):
    import torch.multiprocessing as mp

    world, epochs = 2, 25
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, wl, n_rows, clients, epochs, zipf, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=800)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    dtype = wire.MSG_DTYPE[wire.Workload(wl)]
    mk = (lambda: orc.TatpOracle(n_rows, log_entries=200_000)) if wl == int(W.TATP) else (lambda: orc.SmallbankOracle(n_rows, log_entries=200_000))
    ora = [mk() for _ in range(3)]
    total = 0
    for e in range(epochs):
        for s in range(3):
            parts = [np.frombuffer(res[r][0][e][0][s], dtype) for r in range(world)]
            want = ora[s].replay(np.concatenate(parts))  # the serial order: rank-major concatenation
            lo = 0
            for r in range(world):
                n = len(parts[r])
                assert res[r][0][e][1][s] == want[lo:lo + n].tobytes(), (e, s, r)
                lo += n
                total += n
    assert total > 0 and all(o.errors == 0 for o in ora)
    # final rows: the two ranks' shares together are the oracle's table (as multisets of rows per table)
    for s in range(3):
        for t in e_tables(wl):
            keys = np.concatenate([np.frombuffer(res[r][1][s][t][0], "<u8") for r in range(world)])
            vers = np.concatenate([np.frombuffer(res[r][1][s][t][1], "<u4") for r in range(world)])
            ok, ov, _ = ora[s].dump(t)
            a = sorted(zip(keys.tolist(), vers.tolist()))
            b = sorted(zip(ok.tolist(), ov.tolist()))
            assert a == b, (s, t)
    for r in range(world):
        for st in res[r][2]:
            assert st["foreign_requests"] == 0 and st["route_overflow"] == 0 and st["bad_requests"] == 0
        assert res[r][4]["committed"] > 0
    assert res[0][3] == res[1][3]  # the tightened capacities are agreed by all ranks


# ------------------------------------------------------------------------- RCCL itself, on the one GPU a box has
def _nccl_main(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from dint_amd.driver import Driver
    from dint_amd.replay import Replay, ShardGroup

    n_sub, clients, epochs = 20_000, 6000, 24
    plain = ShardGroup(W.TATP, n_sub, log_entries=200_000)
    routed = ShardGroup(W.TATP, n_sub, log_entries=200_000, transport="nccl", force_exchange=True, n_max=1 << 16)
    assert routed.router.ex.transport == "nccl"
    routed.snapshot()
    d = Driver(W.TATP, clients, n_sub, zipf_theta=0.8)
    ok, trace = True, []
    for _ in range(epochs):  # host path: Router.submit -> pack, dist.all_to_all_single (RCCL), segments, back, unpack
        req = d.next()
        a, b = plain.submit(req), routed.submit(req)
        ok = ok and all(a[s].tobytes() == b[s].tobytes() for s in range(3))
        d.consume(a)
        trace.append((req, a))
    # device path, pipelined over the two buffer sets (what bench.py times): Router.run with the backward half on its
    # own stream next to torch's NCCL stream
    routed.router.tighten_caps()
    routed.restore()
    rp = Replay(trace, routed.msg)
    rp.run(routed, 0, epochs)
    routed.sync()
    rp.check(0, epochs)
    q.put((ok, routed.router.overflow(), d.stats()["committed"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_router_over_rccl_world_size_one():
    """transport "nccl" with one rank: every collective of a step is a real RCCL all_to_all_single on device buffers
    (a self-copy as far as the data goes), on torch's NCCL stream, ordered against the exchange streams and the
    engines' streams by the same events the multi-GPU run uses.  The boxes have one GPU, so this is the branch's only
    execution before the driver's 8-GPU run."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_main, args=(_free_port(), q))
    p.start()
    ok, overflow, committed = q.get(timeout=500)
    p.join(120)
    assert p.exitcode == 0
    assert ok and overflow == 0 and committed > 0


# ------------------------------------------------------------------------- BASELINE configs[4] in its stated shape
def _row_checksum(keys, vers, vals):
    """order-independent checksum of a table share: the sum (mod 2^64) of a 64-bit mix of every row -- the shares of the
    ranks add up to the unsharded table's"""
    k = np.asarray(keys, "<u8").astype(np.uint64)
    h = k * np.uint64(0x9E3779B97F4A7C15) ^ (np.asarray(vers, "<u4").astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F))
    v = np.ascontiguousarray(vals, "u1").reshape(len(k), -1)
    for c in range(0, v.shape[1], 8):
        w = np.zeros(len(k), np.uint64)
        blk = v[:, c:c + 8]
        for b in range(blk.shape[1]):
            w |= blk[:, b].astype(np.uint64) << np.uint64(8 * b)
        h = (h ^ (w + np.uint64(c + 1))) * np.uint64(0xFF51AFD7ED558CCD)
        h ^= h >> np.uint64(33)
    return int(h.sum(dtype=np.uint64)), int(len(k))


def _lock_summary(eng, wl):
    out = []
    for t in e_tables(wl):
        a, b = eng.read_locks(t)
        out.append((int(np.count_nonzero(a)), int(a.sum(dtype=np.uint64)), int(np.count_nonzero(b)), int(b.sum(dtype=np.uint64))))
    return out


def _rank_big(rank, world, port, wl, n_rows, clients, epochs, zipf, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dint_amd.driver import Driver
    from dint_amd.replay import ShardGroup

    torch.cuda.set_device(0)
    grp = ShardGroup(wl, n_rows, device=0, rank=rank, world=world, log_entries=200_000, n_max=1 << 16)
    d = Driver(wl, clients, n_rows, first_client=rank * clients, zipf_theta=zipf)
    trace = []
    for _ in range(epochs):
        req = d.next()
        rep = grp.submit(req)
        d.consume(rep)
        trace.append(([r.tobytes() for r in req], [r.tobytes() for r in rep]))
    sums = [[_row_checksum(*e.dump_rows(t)) for t in e_tables(wl)] for e in grp.engines]
    locks = [_lock_summary(e, wl) for e in grp.engines]
    local = [[int(e.hash_size(t)) for t in e_tables(wl)] for e in grp.engines[:1]]
    q.put((rank, trace, sums, locks, [e.stats() for e in grp.engines], d.stats(), local))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("wl,n_rows,clients,zipf,epochs", [(int(W.SMALLBANK), 80_000_000, 24_000, 0.99 - 1e-9, 6),
                                                           (int(W.TATP), 1_000_001, 16_000, 0.8, 8)])
def test_eight_ranks_at_the_baseline_shape_equal_one_unsharded_server(wl, n_rows, clients, zipf, epochs):
    """BASELINE configs[4] as stated -- SmallBank, 80M accounts hash-sharded over EIGHT ranks (10M accounts each: 3,750,000
    local buckets per table and logical server), accounts ~ Zipf-0.99, closed-loop clients on every rank -- with the eight
    ranks sharing the one GPU of the box (gloo / host transport; the kernels, capacities and orders are those of the 8-GPU
    run).  Every reply byte of every rank must equal what ONE unsharded engine group of 80M accounts answers to the
    rank-major concatenation of the batches (smallbank/udp/smallbank.h:10,16-18, server_shard.cc:75-76), and the ranks'
    table shares must add up to its tables (order-independent row checksums, lock-counter sums).  The TATP case runs the
    same eight ranks at configs[3]'s size + 1: 937,500 and 1,406,251 buckets -- the latter not divisible by 8, so the ranks'
    local tables are ceil-divided and the last rank's share is short (tatp/udp/server_shard.cc:75-79)."""
    import torch.multiprocessing as mp
    from dint_amd.replay import ShardGroup

    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_big, args=(r, world, port, wl, n_rows, clients, epochs, zipf, q)) for r in range(world)]
    for p in procs:
        p.start()
    one = ShardGroup(wl, n_rows, device=0, log_entries=200_000)  # the unsharded servers, populated while the ranks start up
    # ... and the CPU oracle at the SAME table sizes, so that the unsharded engine the ranks are compared with is itself held
    # to the serial restatement at configs[4]'s size.  tatp (1,000,001 subscribers): the whole tables.
    # smallbank (80M accounts = 30,000,000 buckets per table, smallbank/udp/server_shard.cc:75-76): the oracle is sized for
    # 80M accounts but holds only the rows of one bucket in 64 (bucket % 64 == 0: 1.25M accounts per table, loaded with
    # the population's values, smallbank/udp/smallbank.h:105-127); it replays every request that touches those buckets --
    # a request touches one bucket, and its lock slot lock_hash % hash_size is that bucket -- plus every log append.
    SAMPLE = 64
    if wire.Workload(wl) == W.TATP:
        oracles = [orc.TatpOracle(n_rows, log_entries=200_000) for _ in range(3)]
        hs = None
    else:
        oracles = [orc.SmallbankOracle(n_rows, log_entries=200_000, populate_n=0) for _ in range(3)]
        hs = np.uint64(oracles[0].hash_size(0))
        assert int(hs) == n_rows * 3 // 2 // 4 == 30_000_000
        assert int(sd.fasthash_key(np.zeros(1, np.uint64))[0]) == 0x16C38EE185750EBC
        acct = np.arange(n_rows, dtype=np.uint64)
        mine = acct[(sd.fasthash_key(acct) % hs) % np.uint64(SAMPLE) == 0]
        assert 0.9 * n_rows / SAMPLE < len(mine) < 1.1 * n_rows / SAMPLE
        bal = np.frombuffer(np.float32(1e9).tobytes(), np.uint8)
        for t, magic in ((0, 97), (1, 98)):
            vals = np.zeros((len(mine), 8), np.uint8)
            vals[:, 0], vals[:, 4:] = magic, bal
            for o in oracles:
                o.load(t, mine, np.zeros(len(mine), np.uint32), vals)
        del acct
    res = {}
    for _ in range(world):
        r = q.get(timeout=1200)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    dtype = wire.MSG_DTYPE[wire.Workload(wl)]
    total = checked_by_oracle = sampled_table_reqs = 0
    for e in range(epochs):
        for s in range(3):
            parts = [np.frombuffer(res[r][0][e][0][s], dtype) for r in range(world)]
            allreq = np.concatenate(parts)
            want = one.engines[s].submit(allreq)  # the serial order: rank-major concatenation
            if hs is None:
                assert oracles[s].replay(allreq).tobytes() == want.tobytes(), ("oracle", e, s)
                checked_by_oracle += len(allreq)
            else:
                pick = (allreq["type"] == 6) | ((sd.fasthash_key(allreq["key"]) % hs) % np.uint64(SAMPLE) == 0)
                assert oracles[s].replay(allreq[pick].copy()).tobytes() == want[pick].tobytes(), ("oracle", e, s)
                checked_by_oracle += int(pick.sum())
                sampled_table_reqs += int((pick & (allreq["type"] != 6)).sum())
            lo = 0
            for r in range(world):
                n = len(parts[r])
                assert res[r][0][e][1][s] == want[lo:lo + n].tobytes(), (e, s, r)
                lo += n
                total += n
    assert total > world * clients
    # the oracle saw a real share: all of tatp; of smallbank ~1/64 of the cold traffic -- and whatever hot account falls into
    # the sampled buckets -- with table requests among it
    assert checked_by_oracle == total if hs is None else (checked_by_oracle > total // 200 and sampled_table_reqs > 200)
    for s in range(3):
        for ti, t in enumerate(e_tables(wl)):
            cs = sum(res[r][1][s][ti][0] for r in range(world)) % (1 << 64)
            n = sum(res[r][1][s][ti][1] for r in range(world))
            assert (cs, n) == _row_checksum(*one.engines[s].dump_rows(t)), (s, t)
        want_l = _lock_summary(one.engines[s], wl)
        for ti in range(len(want_l)):
            got_l = tuple(sum(res[r][2][s][ti][k] for r in range(world)) for k in range(4))
            assert got_l == want_l[ti], (s, ti)
    for r in range(world):
        for st in res[r][3]:
            assert st["foreign_requests"] == 0 and st["route_overflow"] == 0 and st["bad_requests"] == 0 and st["pool_exhausted"] == 0
        assert res[r][4]["committed"] > 0
    for st in (e.stats() for e in one.engines):
        assert st["bad_requests"] == 0 and st["pool_exhausted"] == 0
    if wire.Workload(wl) == W.TATP:
        assert any(h % world for h in res[0][5][0]), res[0][5]  # a bucket count the ranks cannot divide evenly


@pytest.mark.parametrize("fill", [0.55, 0.8, 1.0])
def test_hot_keys_in_half_filled_segments_stay_in_closed_form(fill):
    """A hot key's pieces are ranges of the request INDEX, and a segmented pass (the closed loop's batches, the slots of the
    exchange) fills only the front of each segment: the pieces are sized for the filled stretches (kv_list_items), so no piece
    holds more requests than a workgroup has threads and no hot key goes the slow way BECAUSE its pass is segmented
    (dint_stats.late_requests: early r06 sent every hot key of a closed-loop pass there, 240 us per epoch).  Against the
    contiguous run of the same requests: same bytes, same rows, no more late requests."""
    T = wire.Tatp
    n, n_sub, nseg = 60_000, 200_000, 3  # (a sparse table: the hot rows have their buckets to themselves, as a rule)
    o = orc.TatpOracle(n_sub, log_entries=200_000, populate_n=3000)
    existing = [o.dump(t)[0] for t in range(5)]
    rng = np.random.default_rng(11)
    req = tracegen.tatp_random(n, existing, seed=12, n_sub_touch=3000)
    u = rng.random(n)
    for key in (5, 77, 1234):  # (the noise leaves the hot rows alone: an INSERT / DELETE of one would take it out of the closed form)
        req["key"][(req["table"] == 0) & (req["key"] == key)] = 2000 + key
    for key, lo, hi in ((5, 0.0, 0.10), (77, 0.10, 0.15), (1234, 0.15, 0.175)):  # 6,000 / 3,000 / 1,500 requests on three subscribers
        hot = (u >= lo) & (u < hi)
        req["table"][hot] = 0
        req["key"][hot] = key
        req["type"][hot] = rng.choice([T.READ, T.ACQUIRE_LOCK, T.ABORT, T.COMMIT_PRIM, T.COMMIT_BCK], int(hot.sum()), p=[0.7, 0.1, 0.04, 0.08, 0.08])
    per = n // nseg
    cuts = [0, per, 2 * per, n]
    cap = int(per / fill) + 8

    def mk():
        e = _engine(W.TATP, n_rows=n_sub, log_entries=200_000)
        e.populate(3000)
        return e
    a, b = mk(), mk()
    want = a.submit(req)
    assert want.tobytes() == o.replay(req).tobytes()
    buf, stride, _ = _segmented(req, cuts, cap)
    d = torch.from_numpy(buf).cuda()
    b.submit_segments(d.data_ptr() + 64, nseg, cap, stride, d.data_ptr(), stride)
    b.sync()
    got = _unsegment(d.cpu().numpy(), cuts, stride, req.dtype)
    assert got.tobytes() == want.tobytes()
    for t in range(5):
        assert all((x == y).all() for x, y in zip(a.dump_rows(t), b.dump_rows(t)))
    st, sa = b.stats(), a.stats()
    assert st["big_bin_requests"] > 9_000 and sa["late_requests"] < 1000  # (the hot rows of the contiguous run are in closed form)
    assert st["late_requests"] <= sa["late_requests"], (st["late_requests"], st["late_items"], sa["late_requests"], cap, nseg * cap)
