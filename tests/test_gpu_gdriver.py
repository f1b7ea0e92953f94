"""The GPU-resident closed-loop driver (csrc/k_txn.hip) against the host driver: the same client
state machines compiled for the device must emit a bit-identical request stream, epoch by epoch, and end with the
same transaction statistics -- each driver closing its own loop through its own three GPU shard servers (the host
driver over dint_submit, the GPU driver without leaving the device)."""
import numpy as np
import pytest

from dint_amd import wire
from dint_amd.driver import Driver, GpuDriver
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
W = wire.Workload


@pytest.mark.parametrize("wl,n_rows,clients,zipf,epochs,fuse", [
    (W.TATP, 20_000, 6000, 0.8, 100, 1), (W.TATP, 3000, 5000, None, 100, 1), (W.TATP, 1_000_000, 70_000, 0.8, 40, 1),
    (W.SMALLBANK, 50_000, 4000, 0.99 - 1e-9, 100, 1), (W.SMALLBANK, 600_000, 5000, None, 100, 1),
    (W.TATP, 20_000, 6000, 0.8, 60, 0), (W.SMALLBANK, 50_000, 4000, 0.99 - 1e-9, 60, 0),
])
def test_gpu_driver_stream_is_bit_identical_to_the_host_driver(wl, n_rows, clients, zipf, epochs, fuse, monkeypatch):
    """fuse = 1: consume on the stream of next() is deferred into the next emit kernel (the batches alternate between
    two buffer sets); fuse = 0 (DINT_TXN_FUSE=0): consume in a kernel of its own"""
    from dint_amd.replay import GpuLoop, ShardGroup

    monkeypatch.setenv("DINT_TXN_FUSE", str(fuse))

    ga = ShardGroup(wl, n_rows, log_entries=400_000)   # served to the host driver
    gb = ShardGroup(wl, n_rows, log_entries=400_000)   # served to the GPU driver
    host = Driver(wl, clients, n_rows, first_client=11, zipf_theta=zipf)
    cap = 4 * clients + 64
    gpu = GpuDriver(wl, clients, n_rows, cap, first_client=11, zipf_theta=zipf)
    loop = GpuLoop(gb, gpu)
    xs = loop.stream.cuda_stream
    for e in range(epochs):
        want = host.next()
        gpu.next(xs)
        loop.stream.synchronize()
        got = gpu.read_batches()
        for s in range(3):
            assert got[s].tobytes() == want[s].tobytes(), (e, s, len(got[s]), len(want[s]))
        host.consume(ga.submit(want))
        for s, eng in enumerate(gb.engines):  # the rest of GpuLoop.epochs(1)
            eng.stream_wait(xs)
            eng.submit_segments(gpu.batch_ptr[s], 1, cap, cap * gb.msg, gpu.counts_ptr + 4 * s, 0)
            eng.stream_signal(xs)
        gpu.consume(xs)
    loop.sync()
    hs, gs = host.stats(), gpu.stats()
    assert gs["overflow"] == 0
    for k in ("txns", "committed", "by_type", "committed_by_type"):
        assert gs[k] == hs[k], k
    assert gs["txns"] > 0 and gs["committed"] > 0
    for s in range(3):
        for t in range(5 if wl == W.TATP else 2):
            assert all((x == y).all() for x, y in zip(ga.engines[s].dump_rows(t), gb.engines[s].dump_rows(t)))


def test_gpu_loop_free_running_matches_oracle_state():
    """GpuLoop.epochs() without any host inspection in between, then the servers' state against CPU oracles that were
    fed the host driver's (identical) stream"""
    from dint_amd.replay import GpuLoop, ShardGroup

    n_sub, clients, epochs = 20_000, 6000, 60
    g = ShardGroup(W.TATP, n_sub, log_entries=400_000)
    gpu = GpuDriver(W.TATP, clients, n_sub, 4 * clients, zipf_theta=0.8)
    loop = GpuLoop(g, gpu)
    loop.epochs(epochs)
    loop.sync()
    host = Driver(W.TATP, clients, n_sub, zipf_theta=0.8)
    ora = [orc.TatpOracle(n_sub, log_entries=400_000) for _ in range(3)]
    for _ in range(epochs):
        req = host.next()
        host.consume([ora[s].replay(req[s]) for s in range(3)])
    for s in range(3):
        for t in range(5):
            assert all((x == y).all() for x, y in zip(g.engines[s].dump_rows(t), ora[s].dump(t))), (s, t)
            lk, _ = g.engines[s].read_locks(t)
            assert (lk == ora[s].locks(t)).all()
        ring, tail = g.engines[s].read_log(400_000)
        assert tail == ora[s].tail and (np.frombuffer(ring.tobytes(), "u1").reshape(-1, 64) == ora[s].ring).all()
        st = g.engines[s].stats()
        assert st["bad_requests"] == 0 and st["missing_keys"] == 0


@pytest.mark.parametrize("wl,n_rows,clients,zipf", [(W.TATP, 20_000, 6000, 0.8), (W.SMALLBANK, 50_000, 4000, 0.99 - 1e-9)])
def test_gpu_loop_through_the_exchange(wl, n_rows, clients, zipf):
    """the GPU-resident clients drive the SHARDED path: pack / all-to-all / in-place passes / all-to-all / unpack with
    every batch size read on the device (dint_route_item.d_n); same statistics and tables as the host driver's run
    through a plain group (tatp/caladan/client_udp_shard.cc:1120-1185 is the loop both restate)"""
    from dint_amd.replay import GpuLoop, ShardGroup

    epochs, cap = 60, 4 * clients + 64
    plain = ShardGroup(wl, n_rows, log_entries=400_000)
    routed = ShardGroup(wl, n_rows, log_entries=400_000, force_exchange=True, n_max=1 << 16)
    host = Driver(wl, clients, n_rows, first_client=3, zipf_theta=zipf)
    for _ in range(epochs):
        host.consume(plain.submit(host.next()))
    gpu = GpuDriver(wl, clients, n_rows, cap, first_client=3, zipf_theta=zipf)
    loop = GpuLoop(routed, gpu)
    loop.epochs(epochs)
    loop.sync()
    hs, gs = host.stats(), gpu.stats()
    assert gs["overflow"] == 0 and routed.router.overflow() == 0
    for k in ("txns", "committed", "by_type", "committed_by_type"):
        assert gs[k] == hs[k], k
    for s in range(3):
        for t in range(5 if wl == W.TATP else 2):
            assert all((x == y).all() for x, y in zip(plain.engines[s].dump_rows(t), routed.engines[s].dump_rows(t)))


@pytest.mark.parametrize("wl,n_rows,clients,zipf", [(W.TATP, 20_000, 6000, 0.8), (W.SMALLBANK, 50_000, 4000, 0.99 - 1e-9)])
def test_gpu_loop_two_client_groups_take_turns(wl, n_rows, clients, zipf):
    """two groups of GPU-resident clients (disjoint client ids) take turns at the servers, each on its own stream, so
    one group's consume / emit kernels overlap the servers' work on the other group's batch; the host run that lets two
    Drivers take turns the same way finishes the same transactions and leaves the same tables"""
    from dint_amd.replay import GpuLoop, ShardGroup

    epochs, cap = 50, 4 * clients + 64
    a, b = ShardGroup(wl, n_rows, log_entries=400_000), ShardGroup(wl, n_rows, log_entries=400_000)
    hosts = [Driver(wl, clients, n_rows, first_client=k * clients, zipf_theta=zipf) for k in range(2)]
    for _ in range(epochs):
        for h in hosts:
            h.consume(a.submit(h.next()))
    gpus = [GpuDriver(wl, clients, n_rows, cap, first_client=k * clients, zipf_theta=zipf) for k in range(2)]
    loop = GpuLoop(b, gpus)
    loop.epochs(epochs)
    loop.sync()
    for h, g in zip(hosts, gpus):
        hs, gs = h.stats(), g.stats()
        assert gs["overflow"] == 0 and hs["txns"] > 0
        for k in ("txns", "committed", "by_type", "committed_by_type"):
            assert gs[k] == hs[k], k
    for s in range(3):
        for t in range(5 if wl == W.TATP else 2):
            assert all((x == y).all() for x, y in zip(a.engines[s].dump_rows(t), b.engines[s].dump_rows(t)))
