"""Hash-sharded multi-GPU front end: one process per GPU, one Engine per rank.

Every request touches exactly one lock slot / bucket, so the key space is partitioned by
``home = global_slot % world`` (the engine stores its share at ``global_slot // world``).
A step on rank r:

  1. home shard of each ingested request (GPU kernel ``dint_home_shard``: same hash/modulus
     as the engine);
  2. stable partition of the batch by home;
  3. all-to-all of the per-destination counts, then of the fixed-size wire messages
     (RCCL over xGMI; the messages are the packed request structs themselves);
  4. the home engine processes what it received, ordered (source rank, original index) --
     i.e. the serial order of the rank-major concatenation of all ingest slices;
  5. inverse all-to-all of the replies, scatter back to the original positions.

The exchange code is device-agnostic torch; tests run it on CPU tensors over gloo with an
injected per-rank server double.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


class ShardedEngine:
    def __init__(self, engine, world: int, rank: int, group=None, msg_size: Optional[int] = None,
                 home_fn: Optional[Callable] = None, local_fn: Optional[Callable] = None):
        """`engine` is this rank's dint_amd.engine.Engine created with shard_index=rank,
        shard_count=world.  `home_fn(req2d) -> uint8[n]` and `local_fn(recv2d) -> None (in place)`
        override the GPU kernels (used by the CPU/gloo tests)."""
        self.engine, self.world, self.rank, self.group = engine, world, rank, group
        self.msg = msg_size if msg_size is not None else engine.msg_size
        self.home_fn, self.local_fn = home_fn, local_fn

    def _home(self, req2d: torch.Tensor, n: int) -> torch.Tensor:
        if self.home_fn is not None:
            return self.home_fn(req2d)
        home = torch.empty(n, dtype=torch.uint8, device=req2d.device)
        st = torch.cuda.current_stream().cuda_stream
        self.engine.home_shard(req2d, n, home, st)
        return home

    def _local(self, recv2d: torch.Tensor, n: int) -> None:
        if self.local_fn is not None:
            self.local_fn(recv2d)
            return
        st = torch.cuda.current_stream().cuda_stream
        self.engine.submit_device(recv2d, n, recv2d, st)

    def submit_device(self, d_req: torch.Tensor, n: int, d_rep: torch.Tensor, splits=None):
        """d_req / d_rep: uint8 tensors of n * msg_size bytes on this rank's device.

        The collective API wants the split sizes as host integers.  Without `splits` they are exchanged and read
        back (two host syncs per call); a caller that already knows them -- a replayed recorded trace, or an
        ingest path with fixed-capacity slots -- passes `splits = (send_splits, recv_splits)` and the whole step
        stays asynchronous on the stream.  Returns the splits used."""
        W, msg = self.world, self.msg
        req2d = d_req.view(n, msg)
        home = self._home(req2d, n).to(torch.int64)
        home = torch.where(home >= W, torch.full_like(home, self.rank), home)  # no home: counted as bad locally
        order = torch.argsort(home, stable=True)
        if splits is None:
            counts = torch.bincount(home, minlength=W)
            recv_counts = torch.empty_like(counts)
            dist.all_to_all_single(recv_counts, counts, group=self.group)
            send_splits = counts.tolist()          # host sync: split sizes must be host integers
            recv_splits = recv_counts.tolist()
        else:
            send_splits, recv_splits = splits
        send = req2d.index_select(0, order).contiguous()
        n_recv = int(sum(recv_splits))
        recv = torch.empty((n_recv, msg), dtype=torch.uint8, device=d_req.device)
        dist.all_to_all_single(recv, send, recv_splits, send_splits, group=self.group)
        if n_recv:
            self._local(recv, n_recv)
        back = torch.empty_like(send)
        dist.all_to_all_single(back, recv, send_splits, recv_splits, group=self.group)
        d_rep.view(n, msg).index_copy_(0, order, back)
        return send_splits, recv_splits
