"""Record / replay of closed-loop transaction traffic against a group of three shard servers.

The reference deployment is three `server_shard <id>` processes (tatp/udp/server_shard.cc:279-285),
3-way replicated.  A :class:`ShardGroup` is those three servers on one GPU (three engines, each on
its own HIP stream so their kernel chains overlap) -- or, with world > 1, on N GPUs: every logical
shard server is hash-partitioned over the ranks and the three batches of an epoch cross the node in one
all-to-all each way (:class:`dint_amd.sharded.Router`).

`record()` runs a :class:`Driver` closed loop through the group and keeps every epoch's per-shard
request and reply batches; `Replay` uploads them to HBM and re-submits them without any host work,
which is what bench.py times (inputs resident in HBM when the timed region starts).
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch

from .driver import N_SHARDS, Driver
from .engine import Engine, submit_segments_multi
from .sharded import Router
from .wire import Workload


class ShardGroup:
    def __init__(self, workload: Workload, n_rows: int, *, device: int = -1, rank: int = 0, world: int = 1,
                 log_entries: int = 0, populate: Optional[int] = None, transport: Optional[str] = None,
                 force_exchange: bool = False, n_max: int = 1 << 20, flags: int = 0):
        self.workload, self.world, self.rank = Workload(workload), world, rank
        self.engines = [Engine(workload, n_rows=n_rows, device=device, shard_index=rank, shard_count=world,
                               log_entries=log_entries, flags=flags) for _ in range(N_SHARDS)]
        self.msg = self.engines[0].msg_size
        self.router = (Router(self.engines, world, rank, transport=transport, n_max=n_max)
                       if world > 1 or force_exchange else None)
        for e in self.engines:  # every server holds every row (tatp/udp/server_shard.cc:71-85)
            e.populate(n_rows if populate is None else populate)

    # ---- host path (recording) ----------------------------------------------------------------------
    def submit(self, reqs: List[np.ndarray]) -> List[np.ndarray]:
        if self.router is None:
            return [self.engines[s].submit(reqs[s]) if len(reqs[s]) else reqs[s] for s in range(N_SHARDS)]
        return self.router.submit(reqs)  # every rank enters the collectives, also with an empty batch

    # ---- device path (replay) --------------------------------------------------------------------------
    def submit_device(self, d_reqs, counts, d_reps, ahead=None) -> None:
        """d_reqs / d_reps: per shard uint8 tensors; asynchronous.  Single GPU: each engine runs on its own
        stream (the three shard servers are independent).  Multi GPU: pack -> all-to-all -> the three engines on
        their own streams -> all-to-all -> unpack, without host round trips (fixed-capacity slots).
        `ahead` = (d_reqs, counts, d_reps) of the NEXT call (single GPU: dint_submit_device_ahead -- the batches of a
        receive ring are in HBM before their turn comes)."""
        if self.router is None:
            for s in range(N_SHARDS):
                if counts[s]:
                    nxt = None if ahead is None else (ahead[0][s], ahead[1][s], ahead[2][s])
                    self.engines[s].submit_device(d_reqs[s], counts[s], d_reps[s], 0, ahead=nxt)
        else:
            self.router.step(d_reqs, counts, d_reps)

    def sync(self):
        if self.router is not None:
            self.router.sync()
        for e in self.engines:
            e.sync()
        torch.cuda.synchronize()

    def snapshot(self):
        for e in self.engines:
            e.snapshot()

    def restore(self):
        for e in self.engines:
            e.restore()


def record(driver: Driver, group, n_epochs: int):
    """Closed loop for n_epochs; returns (trace, finished_txns_per_epoch) with
    trace[e] = (requests[3], replies[3]) as numpy arrays."""
    trace, done = [], []
    last = driver.stats()["txns"]
    for _ in range(n_epochs):
        req = driver.next()
        rep = group.submit(req)
        driver.consume(rep)
        trace.append((req, rep))
        now = driver.stats()["txns"]  # transactions whose last reply arrived in this epoch
        done.append(now - last)
        last = now
    return trace, done


class Replay:
    """A recorded trace resident in HBM: per epoch and shard server the request batch, a reply buffer, and the replies
    of the recorded run (also in HBM: `check` compares on the device).

    `inplace` (round 6): the replay answers every batch IN PLACE, as the reference does (the reply is the request
    struct mutated, tatp/udp/server_shard.cc:116-121) -- the engines then skip the request -> reply copy of their
    partition kernel.  The batches the engines work on are the reply buffers, refilled from the pristine requests by
    `reset()` OUTSIDE any timed region (that refill is the NIC's DMA into the receive ring, not server work).
    `ahead`: every submission announces the next epoch's batch (dint_submit_device_ahead)."""

    def __init__(self, trace=(), msg_size: int = 0, inplace: bool = False, ahead: bool = False):
        self.msg = msg_size
        self.inplace, self.ahead = inplace, ahead
        self.counts, self.d_req, self.d_rep, self.d_want = [], [], [], []
        for req, rep in ((t[0], t[1]) for t in trace):
            self.append(req, rep)

    def append(self, req, rep) -> None:
        def up(a):  # through page-locked staging: torch / HIP never copy pageable memory here (r04: a stale page tail of a
            b = np.frombuffer(a.tobytes(), np.uint8)  # pageable copy under rocprofv3 made every second smallbank profile run fail)
            if len(b) == 0:
                return torch.empty(0, dtype=torch.uint8, device="cuda")
            h = torch.empty(len(b), dtype=torch.uint8).pin_memory()
            h.numpy()[:] = b
            d = h.cuda(non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return d
        self.counts.append([len(req[s]) for s in range(N_SHARDS)])
        self.d_req.append([up(req[s]) for s in range(N_SHARDS)])
        self.d_want.append([up(rep[s]) for s in range(N_SHARDS)])
        self.d_rep.append([torch.empty(len(req[s]) * self.msg, dtype=torch.uint8, device="cuda") for s in range(N_SHARDS)])

    @classmethod
    def recording(cls, driver: Driver, group, n_epochs: int, keep_host: int = 0, inplace: bool = False, ahead: bool = False):
        """Run the closed loop for n_epochs and keep every epoch in HBM; host copies only of the first `keep_host`
        epochs (what the CPU legs replay).  Returns (replay, finished txns per epoch, host trace)."""
        rp, done, host = cls(msg_size=group.msg, inplace=inplace, ahead=ahead), [], []
        last = driver.stats()["txns"]
        for e in range(n_epochs):
            req = driver.next()
            rep = group.submit(req)
            driver.consume(rep)
            rp.append(req, rep)
            if e < keep_host:
                host.append((req, rep))
            now = driver.stats()["txns"]
            done.append(now - last)
            last = now
        return rp, done, host

    def __len__(self):
        return len(self.counts)

    def reset(self, lo: int = 0, hi: int = None) -> None:
        """in-place replay: the pristine request batches into the buffers the engines answer in (device copies)"""
        if not self.inplace:
            return
        for e in range(lo, len(self) if hi is None else hi):
            for s in range(N_SHARDS):
                self.d_rep[e][s].copy_(self.d_req[e][s])
        torch.cuda.current_stream().synchronize()

    def run(self, group: ShardGroup, lo: int, hi: int) -> None:
        if group.router is not None:  # the epochs of a recorded trace are independent batches: pipelined exchange
            group.router.run([(self.d_req[e], self.counts[e], self.d_rep[e]) for e in range(lo, hi)])
            return
        src = self.d_rep if self.inplace else self.d_req
        for e in range(lo, hi):
            nxt = (src[e + 1], self.counts[e + 1], self.d_rep[e + 1]) if self.ahead and e + 1 < hi else None
            group.submit_device(src[e], self.counts[e], self.d_rep[e], ahead=nxt)

    def check(self, lo: int, hi: int) -> None:
        """The replayed replies must equal the recorded ones byte for byte."""
        for e in range(lo, hi):
            for s in range(N_SHARDS):
                if not torch.equal(self.d_rep[e][s], self.d_want[e][s]):
                    got = self.d_rep[e][s].cpu().numpy().reshape(-1, self.msg)
                    want = self.d_want[e][s].cpu().numpy().reshape(-1, self.msg)
                    req = self.d_req[e][s].cpu().numpy().reshape(-1, self.msg)
                    bad = np.nonzero((got != want).any(axis=1))[0]
                    if os.environ.get("DINT_DUMP_DIVERGENCE"):  # the whole batch, for offline analysis (tools/gpu_bisect.sh)
                        np.savez_compressed(os.environ["DINT_DUMP_DIVERGENCE"], req=req, got=got, want=want, epoch=e, shard=s)
                    i = int(bad[0])
                    raise AssertionError(f"replay diverged from the recorded run at epoch {e}, shard {s}: {len(bad)} of {len(got)} "
                                         f"messages differ, first at {i} (last at {int(bad[-1])}): request {req[i].tobytes().hex()} "
                                         f"got {got[i].tobytes().hex()} want {want[i].tobytes().hex()}; differing indices "
                                         f"{bad[:12].tolist()}")

    def ops(self, lo: int, hi: int) -> int:
        return sum(sum(c) for c in self.counts[lo:hi])


class GpuLoop:
    """The closed loop that never leaves the GPU: :class:`dint_amd.driver.GpuDriver` clients and the three shard servers
    of a :class:`ShardGroup`.  One epoch = emit kernel -> the three engines on their own streams, each answering its batch
    in place (the batch size is read on the device) -> consume kernel; no host round trip, no PCIe.  With a router
    (several GPUs, or the exchange forced on) the epoch's batches cross the exchange on the way: pack / unpack read the
    batch sizes on the device as well (dint_route_item.d_n), and the home engines read theirs from the slot headers.

    `gdriver` may be a list of drivers over disjoint client ranges (first_client): the groups take turns at the servers,
    each on its own stream, so one group's consume / emit kernels run while the servers answer the other group's batch
    (the reference's client threads are not in lock step either).  The order of the batches at the servers is fixed --
    group 0, group 1, ..., group 0 -- so a host run that lets :class:`Driver` objects over the same client ranges take
    turns the same way sees the same bytes."""

    def __init__(self, group: ShardGroup, gdriver):
        self.ds = list(gdriver) if isinstance(gdriver, (list, tuple)) else [gdriver]
        for d in self.ds:
            assert d.cap <= min(e.pass_max for e in group.engines)
            if group.router is not None:
                assert d.cap <= group.router.n_max and group.router.multi is not None
        assert group.router is None or len(self.ds) == 1  # the exchange has one set of slots in flight
        self.g, self.d = group, self.ds[0]
        self.streams = [torch.cuda.Stream() for _ in self.ds]
        self.stream = self.streams[0]
        self.msg = group.msg
        # the servers' kernels go on the clients' stream, all engines per launch (dint_submit_segments_multi): an epoch is
        # one stream, no fork / join.  Several groups take turns on a stream each -- the engines order their passes across
        # the groups' streams themselves -- so one group's client kernel runs beside the servers' pass over the other
        # group's batch.  DINT_LOOP_STREAMS=1 forces the per-engine streams (A/B runs).
        self.one_stream = os.environ.get("DINT_LOOP_STREAMS", "0") != "1"

    def epochs(self, n: int) -> None:
        msg, rt = self.msg, self.g.router
        for _ in range(n):
            for d, st in zip(self.ds, self.streams):
                xs, cap = st.cuda_stream, d.cap
                d.next(xs)
                if rt is None and self.one_stream:
                    # the three servers' passes in one set of launches on the clients' stream: no fork / join at all
                    submit_segments_multi(self.g.engines, d.batch_ptr, 1, [cap] * N_SHARDS, cap * msg,
                                          [d.counts_ptr + 4 * s for s in range(N_SHARDS)], 0, xs)
                elif rt is None:
                    for s, e in enumerate(self.g.engines):
                        e.stream_wait(xs)
                        e.submit_segments(d.batch_ptr[s], 1, cap, cap * msg, d.counts_ptr + 4 * s, 0)
                    for e in self.g.engines:  # only now: a signal makes xs wait for its engine, and a stream_wait issued
                        e.stream_signal(xs)   # after it would make the next engine wait for this one (r02 did that)
                else:
                    with torch.cuda.stream(st):  # Router.run waits for the current stream: the emit kernel
                        rt.step(d.batch_ptr, [cap] * N_SHARDS, d.batch_ptr, d_n=[d.counts_ptr + 4 * s for s in range(N_SHARDS)])
                    rt.join(st)
                d.consume(xs)

    def sync(self):
        for st in self.streams:
            st.synchronize()
        self.g.sync()
