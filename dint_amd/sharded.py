"""Hash-sharded multi-GPU front end: one process per GPU, the key space of every logical
server partitioned by ``home = global slot / bucket % world`` (the engine stores its share at ``// world``).

A step on rank r, for S logical servers at once (S = 3 replicated shard servers for tatp / smallbank, 1 for the
lock tables), entirely on the GPU and free of host round trips:

  1. ``dint_route_pack`` (HIP): stable partition of each ingested batch by home rank into fixed-capacity slots
     of ONE exchange buffer -- per peer a chunk ``[counts | slot of server 0 | slot of server 1 | ...]``;
  2. one all-to-all of the equal-size chunks (RCCL over xGMI; ``dist.all_to_all_single`` without split sizes);
  3. every home engine answers the W segments it received, in place, as one serial history ordered
     (source rank, index) -- ``dint_submit_segments`` on the engine's own stream, so the S servers of a rank
     overlap exactly as they do on a single GPU;
  4. the inverse all-to-all;
  5. ``dint_route_unpack`` (HIP): replies back to their original positions.

The order in step 3 is the serial order of the rank-major concatenation of all ingest batches, so results do
not depend on the number of GPUs.  The collective is torch.distributed's; everything else is the C ABI.

Transports: ``"nccl"`` = RCCL on device buffers (one GPU per rank); ``"host"`` = any CPU backend (gloo) with the
chunks staged through host memory -- functional testing with several ranks on ONE GPU, where RCCL refuses to
run.  `Exchange` is the only class that touches torch.distributed.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .wire import Workload

HDR = 64  # bytes of the chunk header: u32 live count of each server's slot


def _align(x: int, a: int) -> int:
    return (x + a - 1) // a * a


class Exchange:
    """all-to-all of equal-size per-peer chunks of a uint8 device tensor."""

    def __init__(self, world: int, rank: int, group=None, transport: Optional[str] = None):
        self.world, self.rank, self.group = world, rank, group
        if transport is None:
            transport = "nccl" if dist.is_initialized() and dist.get_backend(group) == "nccl" else "host"
        if world == 1 and not dist.is_initialized():
            transport = "self"
        self.transport = transport

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        """out chunk k <- rank k's inp chunk `rank`; enqueued on torch's current stream."""
        if self.transport == "self":
            out.copy_(inp, non_blocking=True)
        elif self.transport == "nccl":
            dist.all_to_all_single(out, inp, group=self.group)
        else:  # staged through the host: synchronous
            h_in = inp.cpu()
            h_out = torch.empty_like(h_in)
            dist.all_to_all_single(h_out, h_in, group=self.group)
            out.copy_(h_out)

    def max_int(self, v: Sequence[int]) -> List[int]:
        if self.transport == "self":
            return list(v)
        dev = "cuda" if self.transport == "nccl" else "cpu"
        t = torch.tensor(list(v), dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return [int(x) for x in t.tolist()]


class Router:
    """The exchange around S engines of this rank (each created with shard_index = rank, shard_count = world)."""

    def __init__(self, engines, world: int, rank: int, *, group=None, transport: Optional[str] = None,
                 n_max: int = 1 << 20, caps: Optional[Sequence[int]] = None, device: str = "cuda"):
        """device="cpu" (with engine doubles that implement the routing calls on host memory) is what the
        world-size-2 gloo tests run; the orchestration below is the same."""
        self.engines, self.world, self.rank, self.device = list(engines), world, rank, device
        self.S = len(self.engines)
        assert 1 <= self.S <= HDR // 4
        self.msg = self.engines[0].msg_size
        self.ex = Exchange(world, rank, group, transport)
        self.n_max = n_max
        self.stream = torch.cuda.Stream() if device == "cuda" else None  # routing kernels and collectives
        # the engines' own streams, as torch streams (event waits / records only; kernels are launched by the ABI)
        self.estream = [torch.cuda.ExternalStream(e.stream) for e in self.engines] if device == "cuda" else None
        # the backward half of a step (inverse all-to-all + unpack) on a stream of its own: the exchange stream then
        # carries only pack + forward all-to-all, and the two halves of consecutive steps run side by side
        self.bstream = (torch.cuda.Stream() if device == "cuda" and os.environ.get("DINT_BWD_STREAM", "1") != "0" else None)
        self.NBUF = 3  # exchange buffer sets: step k uses set k % 3 (see run(): the forward exchange runs TWO steps ahead)
        self.ev_unpacked = [None] * self.NBUF  # per buffer set: its last unpack has read the send buffer
        self.multi = None  # (pack, unpack) of several batches per launch set: the real engines on a GPU
        if device == "cuda" and self.S <= 4:
            from .engine import route_pack_multi, route_unpack_multi
            self.multi = (route_pack_multi, route_unpack_multi)
        self.d_slot = [[torch.empty(n_max, dtype=torch.int32, device=device) for _ in range(self.S)] for _ in range(self.NBUF)]
        self.send = self.recv = None
        self.max_seen = [0] * self.S
        self.set_caps(caps if caps is not None else [self.default_cap(n_max)] * self.S)

    def default_cap(self, n: int) -> int:
        """slot capacity for batches of up to n requests: 1.5x the mean per destination + slack (hash-partitioned
        keys are balanced up to the hot keys), never more than one kernel pass takes"""
        cap = _align(max(1024, (3 * n) // (2 * self.world) + 512), 64) if self.world > 1 else _align(max(n, 64), 64)
        return max(64, min(cap, self.engines[0].pass_max // 64 * 64))

    def set_caps(self, caps: Sequence[int]) -> None:
        """(re)build the exchange buffers for per-server slot capacities `caps` (identical on every rank)"""
        self.caps = [max(2, int(c)) for c in caps]
        self.off = []
        o = HDR
        for c in self.caps:
            self.off.append(o)
            o += _align(c * self.msg, 16)
        self.chunk = _align(o, 64)
        if self.device == "cuda":
            torch.cuda.synchronize()
        self.ev_unpacked = [None] * self.NBUF
        self.send = [torch.zeros(self.world * self.chunk, dtype=torch.uint8, device=self.device) for _ in range(self.NBUF)]
        self.recv = [torch.zeros(self.world * self.chunk, dtype=torch.uint8, device=self.device) for _ in range(self.NBUF)]

    def tighten_caps(self, slack: int = 64) -> List[int]:
        """after a recorded run: the smallest capacities that held every slot seen so far, agreed by all ranks"""
        caps = [_align(m + slack, 64) for m in self.ex.max_int(self.max_seen)]
        self.set_caps(caps)
        return caps

    # ---- the exchange ------------------------------------------------------------------------------------------
    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _forward(self, k: int, d_reqs, counts, track: bool, d_n=None):
        """pack + all-to-all of step k into buffer set k % 2; returns the event the engines wait for.  d_n: per server a
        device pointer to the batch's live request count (counts[s] is then only its upper bound)"""
        b = k % self.NBUF
        xs = self.stream.cuda_stream if self.stream is not None else 0
        sp = self.send[b].data_ptr()
        assert all(counts[s] <= self.n_max for s in range(self.S))
        if self.bstream is not None and self.ev_unpacked[b] is not None:
            self.stream.wait_event(self.ev_unpacked[b])  # step k - 2 is done with this buffer set
        if self.multi:  # the S servers' batches in one set of launches (grid.y = server)
            self.multi[0](self.engines, d_reqs, counts, [sp + self.off[s] for s in range(self.S)], self.caps, self.chunk,
                          [sp + 4 * s for s in range(self.S)], self.chunk, self.d_slot[b], xs, d_n)
        else:
            assert d_n is None, "device-side batch sizes need the multi-batch routing calls"
            for s, e in enumerate(self.engines):
                e.route_pack(d_reqs[s], counts[s], sp + self.off[s], self.caps[s], self.chunk, sp + 4 * s, self.chunk,
                             self.d_slot[b][s], xs)
        if track:  # slot occupancy (host sync; recording runs only)
            hdr = self.send[b].view(self.world, self.chunk)[:, :4 * self.S].cpu().numpy().view("<u4")
            for s in range(self.S):
                self.max_seen[s] = max(self.max_seen[s], int(hdr[:, s].max()))
        self.ex.all_to_all(self.recv[b], self.send[b])
        if self.stream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def _engines(self, k: int, ev_fwd, ev_next=None):
        """every home engine answers its W segments of step k on its own stream; returns their completion events.
        ev_next: the forward exchange of step k + 1 (buffer set (k + 1) % NBUF) -- its segments are ANNOUNCED to the engines
        (dint_submit_segments_multi_ahead: their partition stage rides in this step's launch set)"""
        b = k % self.NBUF
        rp = self.recv[b].data_ptr()
        done = []
        if self._one_set():  # all home engines in one set of launches on the first engine's stream
            from .engine import submit_segments_multi
            self.estream[0].wait_event(ev_fwd)
            ahead = None
            if ev_next is not None and os.environ.get("DINT_ROUTER_NO_AHEAD", "0") != "1":
                self.estream[0].wait_event(ev_next)
                rn = self.recv[(k + 1) % self.NBUF].data_ptr()
                ahead = ([rn + self.off[s] for s in range(self.S)], [rn + 4 * s for s in range(self.S)])
            submit_segments_multi(self.engines, [rp + self.off[s] for s in range(self.S)], self.world, self.caps, self.chunk,
                                  [rp + 4 * s for s in range(self.S)], self.chunk, self.estream[0].cuda_stream, ahead=ahead)
            ev = torch.cuda.Event()
            ev.record(self.estream[0])
            return [ev]
        for s, e in enumerate(self.engines):
            if self.estream is not None:
                self.estream[s].wait_event(ev_fwd)
            e.submit_segments(rp + self.off[s], self.world, self.caps[s], self.chunk, rp + 4 * s, self.chunk)
            if self.estream is not None:
                ev = torch.cuda.Event()
                ev.record(self.estream[s])
                done.append(ev)
        return done

    def _one_set(self) -> bool:
        """the S home engines' passes as one launch set (dint_submit_segments_multi): kv engines whose W segments fit one
        kernel pass.  DINT_ROUTER_STREAMS=1 keeps one stream per engine (A/B runs)."""
        if self.estream is None or self.S < 2 or self.S > 4 or os.environ.get("DINT_ROUTER_STREAMS", "0") == "1":
            return False
        kv = (int(Workload.STORE), int(Workload.TATP), int(Workload.SMALLBANK))
        return all(int(getattr(e, "workload", -1)) in kv and self.world * c <= e.pass_max for e, c in zip(self.engines, self.caps))

    def _backward(self, k: int, ev_done, d_reqs, counts, d_reps, d_n=None):
        b = k % self.NBUF
        st = self.bstream if self.bstream is not None else self.stream
        xs = st.cuda_stream if st is not None else 0
        for ev in ev_done:
            st.wait_event(ev)
        with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
            self.ex.all_to_all(self.send[b], self.recv[b])
        sp = self.send[b].data_ptr()
        if self.multi:
            self.multi[1](self.engines, [sp + self.off[s] for s in range(self.S)], self.caps, self.chunk, self.d_slot[b],
                          d_reqs, counts, d_reps, xs, d_n)
        else:
            for s, e in enumerate(self.engines):
                e.route_unpack(sp + self.off[s], self.caps[s], self.chunk, self.d_slot[b][s], d_reqs[s], counts[s],
                               d_reps[s], xs)
        if self.bstream is not None:
            ev = torch.cuda.Event()
            ev.record(st)
            self.ev_unpacked[b] = ev

    def run(self, steps, track: bool = False) -> None:
        """steps: [(d_reqs[S], counts[S], d_reps[S][, d_n[S]])] -- independent batches (a recorded trace).  Asynchronous.

        Software pipeline over three buffer sets (r06; two until r05): the forward exchange runs TWO steps ahead -- that of step
        k+2 is issued before the backward exchange of step k -- so that when the engines start on step k the segments of step
        k+1 have arrived and can be announced to them (their partition stage then rides in step k's launch set), step k+2's
        requests travel while the engines work on step k+1, and step k's replies travel beside.  Every rank issues the same
        sequence, so the collectives match up."""
        if not steps:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())  # the caller's uploads of the request tensors
            if self.bstream is not None:
                self.bstream.wait_stream(torch.cuda.current_stream())
        with self._on_stream():
            dn = lambda k: steps[k][3] if len(steps[k]) > 3 else None  # noqa: E731
            n = len(steps)
            ev = [None] * (n + 2)
            ev[0] = self._forward(0, steps[0][0], steps[0][1], track, dn(0))
            if n > 1:
                ev[1] = self._forward(1, steps[1][0], steps[1][1], track, dn(1))
            done = self._engines(0, ev[0], ev[1])
            for k in range(n):
                if k + 2 < n:
                    ev[k + 2] = self._forward(k + 2, steps[k + 2][0], steps[k + 2][1], track, dn(k + 2))
                self._backward(k, done, *steps[k][:3], dn(k))
                if k + 1 < n:
                    done = self._engines(k + 1, ev[k + 1], ev[k + 2])

    def step(self, d_reqs, counts, d_reps, track: bool = False, d_n=None) -> None:
        """one batch per server: d_reqs / d_reps are per server uint8 device tensors (or raw device pointers) of
        counts[s] messages -- or, with d_n (device pointers to the live counts), of at most counts[s]"""
        self.run([(d_reqs, counts, d_reps) if d_n is None else (d_reqs, counts, d_reps, d_n)], track)

    def join(self, stream) -> None:
        """`stream` (a torch stream) waits for everything this router has enqueued: the replies of the last step are
        in place"""
        if self.stream is not None:
            stream.wait_stream(self.stream)
        if self.bstream is not None:
            stream.wait_stream(self.bstream)

    def sync(self) -> None:
        if self.stream is not None:
            self.stream.synchronize()
        if self.bstream is not None:
            self.bstream.synchronize()
        for e in self.engines:
            e.sync()

    def overflow(self) -> int:
        return sum(e.stats()["route_overflow"] for e in self.engines)

    # ---- host convenience (recording, tests) ---------------------------------------------------------------------
    def grow_caps(self) -> bool:
        """After a step whose slots overflowed (the refused requests were answered "send again"): a slot that was
        full -- its header count is clamped to the capacity, so the true demand is unknown -- doubles, agreed by all
        ranks.  Returns whether anything changed.  (max_seen is tracked by `submit`; a free-running caller reads
        `overflow()` at its sync points and calls this.)"""
        full = self.ex.max_int([int(m >= c) for m, c in zip(self.max_seen, self.caps)])
        lim = self.engines[0].pass_max // 64 * 64
        new = [min(max(self.n_max, c), lim, 2 * c) if f else c for c, f in zip(self.caps, full)]
        if new == self.caps:
            return False
        self.set_caps(new)
        return True

    def submit(self, reqs: List[np.ndarray], on_overflow: str = "raise") -> List[np.ndarray]:
        """one step from host arrays.  A full slot: "raise" (recording runs must be lossless) or "refuse" -- the
        requests that did not fit come back with the back-pressure reply (their senders send them again) and the
        capacities grow for the next step."""
        d_req = [torch.from_numpy(np.frombuffer(r.tobytes(), np.uint8).copy()).to(self.device) for r in reqs]
        d_rep = [torch.empty_like(d) for d in d_req]
        before = self.overflow()
        self.step(d_req, [len(r) for r in reqs], d_rep, track=True)
        self.sync()
        # route_overflow is per sender: the ranks first AGREE on whether any slot filled (a rank that raised before the
        # collective would leave the others waiting in it for ever -- ADVICE r03), then all raise or all grow
        over = bool(self.ex.max_int([int(self.overflow() != before)])[0])
        if over and on_overflow == "raise":
            raise RuntimeError("exchange slot overflow on some rank: raise the slot capacities (Router.set_caps)")
        out = [np.frombuffer(d.cpu().numpy().tobytes(), r.dtype) for d, r in zip(d_rep, reqs)]
        if over:  # every rank rebuilds its buffers together
            self.grow_caps()
        return out
