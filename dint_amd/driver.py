"""ctypes binding of the closed-loop transaction drivers (include/dint_driver.h, csrc/txn_driver.cc).

A :class:`Driver` plays W reference clients in lock step: ``next()`` returns the three per-shard
request batches of one epoch (numpy arrays of packed wire structs), ``consume(replies)`` feeds the
servers' replies back.  :func:`run_epochs` wires a driver to three servers (GPU engines, or any
object with ``submit``).
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib
from .wire import MSG_DTYPE, Workload

N_SHARDS = 3


class DriverConfig(C.Structure):
    _fields_ = [("workload", C.c_uint32), ("n_clients", C.c_uint32), ("n_rows", C.c_uint64),
                ("first_client", C.c_uint32), ("key_dist", C.c_uint32), ("zipf_theta", C.c_double),
                ("reserved", C.c_uint32 * 8)]


class DriverStats(C.Structure):
    _fields_ = [("txns", C.c_uint64), ("committed", C.c_uint64), ("messages", C.c_uint64),
                ("by_type", C.c_uint64 * 8), ("committed_by_type", C.c_uint64 * 8), ("epochs", C.c_uint64)]


def _bind():
    L = _lib.load()
    if getattr(L, "_driver_bound", False):
        return L
    vp, u32 = C.c_void_p, C.c_uint32
    L.dint_driver_create.restype, L.dint_driver_create.argtypes = C.c_int, [C.POINTER(DriverConfig), C.POINTER(vp)]
    L.dint_driver_destroy.restype, L.dint_driver_destroy.argtypes = None, [vp]
    L.dint_driver_msg_size.restype, L.dint_driver_msg_size.argtypes = C.c_int, [vp]
    L.dint_driver_next.restype, L.dint_driver_next.argtypes = C.c_int, [vp, C.POINTER(u32 * N_SHARDS)]
    L.dint_driver_batch.restype, L.dint_driver_batch.argtypes = vp, [vp, u32]
    L.dint_driver_consume.restype, L.dint_driver_consume.argtypes = C.c_int, [vp, C.POINTER(vp * N_SHARDS)]
    L.dint_driver_get_stats.restype, L.dint_driver_get_stats.argtypes = C.c_int, [vp, C.POINTER(DriverStats)]
    L._driver_bound = True
    return L


class Driver:
    def __init__(self, workload: Workload, n_clients: int, n_rows: int, *, first_client: int = 0,
                 zipf_theta: float | None = None):
        self._L = _bind()
        self.workload = Workload(workload)
        self.dtype = MSG_DTYPE[self.workload]
        cfg = DriverConfig(workload=int(workload), n_clients=n_clients, n_rows=n_rows, first_client=first_client,
                           key_dist=0 if zipf_theta is None else 1, zipf_theta=zipf_theta or 0.0)
        h = C.c_void_p()
        rc = self._L.dint_driver_create(C.byref(cfg), C.byref(h))
        if rc:
            raise _lib.DintError(f"dint_driver_create failed: {rc}")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.dint_driver_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter exit the HIP runtime may already be gone (module teardown order): the process is ending anyway
        if not sys.is_finalizing():
            self.close()

    def next(self):
        """Requests of the next epoch: list of 3 numpy arrays (copies) of packed wire structs."""
        cnt = (C.c_uint32 * N_SHARDS)()
        _lib.check(self._L.dint_driver_next(self._h, C.byref(cnt)))
        out = []
        for s in range(N_SHARDS):
            n = cnt[s]
            if n == 0:
                out.append(np.zeros(0, self.dtype))
                continue
            p = self._L.dint_driver_batch(self._h, s)
            buf = (C.c_uint8 * (n * self.dtype.itemsize)).from_address(p)
            out.append(np.frombuffer(buf, self.dtype, n).copy())
        return out

    def consume(self, replies):
        keep = [np.ascontiguousarray(r) for r in replies]
        ptrs = (C.c_void_p * N_SHARDS)(*[r.ctypes.data if len(r) else None for r in keep])
        _lib.check(self._L.dint_driver_consume(self._h, C.byref(ptrs)))

    def stats(self) -> dict:
        s = DriverStats()
        _lib.check(self._L.dint_driver_get_stats(self._h, C.byref(s)))
        return {"txns": s.txns, "committed": s.committed, "messages": s.messages, "epochs": s.epochs,
                "by_type": list(s.by_type), "committed_by_type": list(s.committed_by_type)}


class GpuDriver:
    """The same closed-loop clients resident on the GPU (csrc/k_txn.hip, include/dint_driver.h dint_gdriver_*):
    ``next(stream)`` emits the epoch's three request batches into device arrays (``batch_ptr[s]``, refreshed by every
    ``next``; live sizes at ``counts_ptr``), the shard servers answer in place, ``consume(stream)`` feeds the replies
    back (on the stream of ``next``: fused into the next emit kernel).  The request stream is bit-identical to
    :class:`Driver`'s."""

    def __init__(self, workload: Workload, n_clients: int, n_rows: int, cap: int, *, first_client: int = 0,
                 zipf_theta: float | None = None, device: int = -1):
        L = self._L = _lib.load()
        vp, u32 = C.c_void_p, C.c_uint32
        L.dint_gdriver_create.restype, L.dint_gdriver_create.argtypes = C.c_int, [C.POINTER(DriverConfig), C.c_int32, u32, C.POINTER(vp)]
        L.dint_gdriver_destroy.restype, L.dint_gdriver_destroy.argtypes = None, [vp]
        L.dint_gdriver_next.restype, L.dint_gdriver_next.argtypes = C.c_int, [vp, vp]
        L.dint_gdriver_consume.restype, L.dint_gdriver_consume.argtypes = C.c_int, [vp, vp]
        L.dint_gdriver_batch.restype, L.dint_gdriver_batch.argtypes = vp, [vp, u32]
        L.dint_gdriver_counts.restype, L.dint_gdriver_counts.argtypes = vp, [vp]
        L.dint_gdriver_get_stats.restype, L.dint_gdriver_get_stats.argtypes = C.c_int, [vp, C.POINTER(DriverStats), C.POINTER(C.c_uint64)]
        L.dint_gdriver_read_batch.restype, L.dint_gdriver_read_batch.argtypes = C.c_int64, [vp, u32, vp, C.c_uint64]
        self.workload = Workload(workload)
        self.dtype = MSG_DTYPE[self.workload]
        self.cap = cap
        cfg = DriverConfig(workload=int(workload), n_clients=n_clients, n_rows=n_rows, first_client=first_client,
                           key_dist=0 if zipf_theta is None else 1, zipf_theta=zipf_theta or 0.0)
        h = vp()
        rc = L.dint_gdriver_create(C.byref(cfg), device, cap, C.byref(h))
        if rc:
            raise _lib.DintError(f"dint_gdriver_create failed: {rc}")
        self._h = h
        self.batch_ptr = [L.dint_gdriver_batch(h, s) for s in range(N_SHARDS)]
        self.counts_ptr = L.dint_gdriver_counts(h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.dint_gdriver_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter exit the HIP runtime may already be gone (module teardown order): the process is ending anyway
        if not sys.is_finalizing():
            self.close()

    def next(self, stream: int = 0):
        rc = self._L.dint_gdriver_next(self._h, stream)
        if rc:
            raise _lib.DintError(f"dint_gdriver_next failed: {rc}")
        # the epoch's batches alternate between two buffer sets (replies of epoch k are read while epoch k+1 is written)
        self.batch_ptr = [self._L.dint_gdriver_batch(self._h, s) for s in range(N_SHARDS)]

    def consume(self, stream: int = 0):
        rc = self._L.dint_gdriver_consume(self._h, stream)
        if rc:
            raise _lib.DintError(f"dint_gdriver_consume failed: {rc}")

    def stats(self) -> dict:
        s, ov = DriverStats(), C.c_uint64()
        rc = self._L.dint_gdriver_get_stats(self._h, C.byref(s), C.byref(ov))
        if rc:
            raise _lib.DintError(f"dint_gdriver_get_stats failed: {rc}")
        return {"txns": s.txns, "committed": s.committed, "messages": s.messages, "epochs": s.epochs,
                "by_type": list(s.by_type), "committed_by_type": list(s.committed_by_type), "overflow": ov.value}

    def read_batches(self):
        """host copies of the current epoch's three batches (synchronises; tests and debugging)"""
        out = []
        for s in range(N_SHARDS):
            buf = np.zeros(self.cap, self.dtype)
            n = self._L.dint_gdriver_read_batch(self._h, s, buf.ctypes.data, self.cap)
            if n < 0:
                raise _lib.DintError(f"dint_gdriver_read_batch failed: {n}")
            out.append(buf[:n].copy())
        return out


class FasstClientConfig(C.Structure):
    _fields_ = [("n_workers", C.c_uint32), ("first_worker", C.c_uint32), ("key_space", C.c_uint32),
                ("read_pct", C.c_uint32), ("key_dist", C.c_uint32), ("reserved0", C.c_uint32),
                ("zipf_theta", C.c_double), ("reserved", C.c_uint32 * 8)]


class FasstClientStats(C.Structure):
    _fields_ = [("requests", C.c_uint64), ("epochs", C.c_uint64), ("committed", C.c_uint64), ("rejects", C.c_uint64),
                ("rollbacks", C.c_uint64), ("protocol_errors", C.c_uint64), ("reserved", C.c_uint64 * 2)]


class FasstClient:
    """The lock_fasst load generator (lock_fasst/caladan/client.cc:183-280 restated, csrc/fasst_client.cc): W
    workers in lock step, ``next()`` = one 9-byte request per worker, ``consume(replies)`` advances them."""

    def __init__(self, n_workers: int = 4096, key_space: int = 24_000_000, *, read_pct: int = 80,
                 zipf_theta: float | None = 0.8, first_worker: int = 0):
        from .wire import FASST_MSG

        L = self._L = _lib.load()
        vp = C.c_void_p
        L.dint_fasst_client_create.restype, L.dint_fasst_client_create.argtypes = C.c_int, [C.POINTER(FasstClientConfig), C.POINTER(vp)]
        L.dint_fasst_client_destroy.restype, L.dint_fasst_client_destroy.argtypes = None, [vp]
        L.dint_fasst_client_next.restype, L.dint_fasst_client_next.argtypes = vp, [vp]
        L.dint_fasst_client_consume.restype, L.dint_fasst_client_consume.argtypes = C.c_int, [vp, vp]
        L.dint_fasst_client_get_stats.restype, L.dint_fasst_client_get_stats.argtypes = C.c_int, [vp, C.POINTER(FasstClientStats)]
        self.dtype, self.n = FASST_MSG, n_workers
        cfg = FasstClientConfig(n_workers=n_workers, first_worker=first_worker, key_space=key_space, read_pct=read_pct,
                                key_dist=0 if zipf_theta is None else 1, zipf_theta=zipf_theta or 0.0)
        h = vp()
        rc = L.dint_fasst_client_create(C.byref(cfg), C.byref(h))
        if rc:
            raise _lib.DintError(f"dint_fasst_client_create failed: {rc}")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.dint_fasst_client_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter exit the HIP runtime may already be gone (module teardown order): the process is ending anyway
        if not sys.is_finalizing():
            self.close()

    def next(self) -> np.ndarray:
        p = self._L.dint_fasst_client_next(self._h)
        if not p:
            raise _lib.DintError("dint_fasst_client_next: replies of the previous epoch are still outstanding")
        return np.frombuffer((C.c_uint8 * (self.n * 9)).from_address(p), self.dtype).copy()

    def consume(self, replies: np.ndarray):
        replies = np.ascontiguousarray(replies)
        _lib.check(self._L.dint_fasst_client_consume(self._h, replies.ctypes.data))

    def peek(self, worker: int = 0):
        """(read set, write set) of the transaction `worker` is running"""
        k, w = (C.c_uint32 * 10)(), (C.c_uint32 * 10)()
        nk, nw = C.c_uint32(), C.c_uint32()
        self._L.dint_fasst_client_peek.restype = C.c_int
        self._L.dint_fasst_client_peek.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.check(self._L.dint_fasst_client_peek(self._h, worker, k, C.byref(nk), w, C.byref(nw)))
        return list(k[:nk.value]), list(w[:nw.value])

    def stats(self) -> dict:
        s = FasstClientStats()
        _lib.check(self._L.dint_fasst_client_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_ if k != "reserved"}


def fasst_trace(server, n_requests: int = 24_000_000, **kw):
    """The lock_fasst request trace: closed loop of a FasstClient against `server` (an object with
    submit(ndarray) -> ndarray) until n_requests requests were issued.  Returns (requests, replies, client stats);
    the trace is a function of the client parameters alone as long as the server is correct."""
    c = FasstClient(**kw)
    reqs, reps, n = [], [], 0
    while n < n_requests:
        r = c.next()[:n_requests - n]  # the last epoch is cut: the server sees exactly n_requests requests
        p = server.submit(r)
        if len(r) == c.n:
            c.consume(p)
        reqs.append(r)
        reps.append(p)
        n += len(r)
    st = c.stats()
    return np.concatenate(reqs), np.concatenate(reps), st


class TplClient:
    """The lock_2pl load generator, W workers in lock step (numpy over the workers): lock_2pl/caladan/client.cc:167-240
    over transactions shaped by lock_2pl/caladan/trace_init.sh:6-27 -- 5..10 distinct locks in ascending order, each
    exclusive with probability 1 - read_prop -- acquire them one by one; a REJECT releases what the transaction holds
    (in acquisition order) and starts it again; once all are held, release them in reverse order.  ``next()`` = one
    6-byte request per worker, ``consume(replies)`` advances them.  Only granted locks are ever released, as in the
    reference."""

    ACQ, ROLL, REL = 0, 1, 2

    def __init__(self, n_workers: int = 4096, key_space: int = 24_000_000, *, read_prop: float = 0.8,
                 zipf_theta: float | None = 0.8, seed: int = 0xDEADBEEF):
        from .wire import TPL_MSG
        from .workloads import Zipf

        self.dtype, self.n, self.key_space, self.read_prop = TPL_MSG, n_workers, key_space, read_prop
        self.rng = np.random.default_rng(seed)
        self.z = Zipf(key_space, zipf_theta or 0.0, seed + 1)
        W = n_workers
        self.lid = np.zeros((W, 10), np.uint32)
        self.typ = np.zeros((W, 10), np.uint8)
        self.nlock = np.zeros(W, np.int64)
        self.mode = np.zeros(W, np.int64)
        self.pos = np.zeros(W, np.int64)   # ACQ: lock being acquired; ROLL: lock being released; REL: lock being released
        self.held = np.zeros(W, np.int64)  # locks the transaction holds (ROLL releases 0 .. held-1)
        self.stats_ = {"requests": 0, "committed": 0, "rejects": 0, "protocol_errors": 0}
        self._new_txn(np.arange(W))
        self._out = False

    def _new_txn(self, w):
        if len(w) == 0:
            return
        n = self.rng.integers(5, 11, len(w))
        k = self.z.sample((len(w), 10)).astype(np.uint32)
        k[np.arange(10)[None, :] >= n[:, None]] = 0xFFFFFFFF  # unused tail sorts last
        for _ in range(64):  # distinct lids per transaction (random.sample in trace_init.sh)
            k.sort(axis=1)
            dup = (k[:, 1:] == k[:, :-1]) & (k[:, 1:] != 0xFFFFFFFF)
            if not dup.any():
                break
            r, c = np.nonzero(dup)
            k[r, c + 1] = self.rng.integers(0, self.key_space, len(r))
        self.lid[w] = k
        self.typ[w] = (self.rng.random((len(w), 10)) >= self.read_prop).astype(np.uint8)
        self.nlock[w], self.mode[w], self.pos[w], self.held[w] = n, self.ACQ, 0, 0

    def next(self) -> np.ndarray:
        assert not self._out, "replies of the previous epoch are still outstanding"
        m = np.zeros(self.n, self.dtype)
        r = np.arange(self.n)
        m["action"] = (self.mode != self.ACQ).astype(np.uint8)  # 0 ACQUIRE, 1 RELEASE (lock_2pl/udp/net.h:11-16)
        m["lid"] = self.lid[r, self.pos]
        m["type"] = self.typ[r, self.pos]
        self._out = True
        self.stats_["requests"] += self.n
        return m

    def consume(self, rep: np.ndarray):
        a = rep["action"]
        acq, roll, rel = self.mode == self.ACQ, self.mode == self.ROLL, self.mode == self.REL
        self.stats_["protocol_errors"] += int((acq & ~np.isin(a, (2, 3))).sum() + ((roll | rel) & (a != 5)).sum())
        grant, rej = acq & (a == 2), acq & (a == 3)
        self.stats_["rejects"] += int(rej.sum())
        self.pos[grant] += 1
        self.held[grant] += 1
        full = grant & (self.pos == self.nlock)
        self.mode[full], self.pos[full] = self.REL, self.nlock[full] - 1
        rb = rej & (self.held > 0)          # release what is held, then start the transaction again
        self.mode[rb], self.pos[rb] = self.ROLL, 0
        # (a reject with nothing held: the same ACQUIRE goes out again)
        self.pos[roll] += 1
        done_roll = roll & (self.pos == self.held)
        self.mode[done_roll], self.pos[done_roll], self.held[done_roll] = self.ACQ, 0, 0
        self.pos[rel] -= 1
        fin = rel & (self.pos < 0)
        self.stats_["committed"] += int(fin.sum())
        self._new_txn(np.nonzero(fin)[0])
        self._out = False

    def stats(self) -> dict:
        return dict(self.stats_)


def tpl_trace(server, n_requests: int, **kw):
    """closed loop of a TplClient against `server` (submit(ndarray) -> ndarray) for n_requests requests"""
    c = TplClient(**kw)
    reqs, reps, n = [], [], 0
    while n < n_requests:
        r = c.next()[:n_requests - n]
        p = server.submit(r)
        if len(r) == c.n:
            c.consume(p)
        reqs.append(r)
        reps.append(p)
        n += len(r)
    return np.concatenate(reqs), np.concatenate(reps), c.stats()


def run_epochs(driver: Driver, servers, n_epochs: int, record: bool = False):
    """Closed loop: `servers` = 3 objects with submit(ndarray) -> ndarray (one per shard).
    Returns the recorded [(requests[3], replies[3])] per epoch when record=True."""
    trace = []
    for _ in range(n_epochs):
        req = driver.next()
        rep = [servers[s].submit(req[s]) if len(req[s]) else req[s] for s in range(N_SHARDS)]
        driver.consume(rep)
        if record:
            trace.append((req, rep))
    return trace
