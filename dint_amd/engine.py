"""Host-side mirror of the reference servers over the C ABI.

One :class:`Engine` plays the role of one reference server process
(``lock_fasst/udp/server``, ``tatp/udp/server_shard`` ...): it owns the tables and
answers batches of wire messages.  ``submit`` takes/returns numpy arrays of the packed
wire structs (:mod:`dint_amd.wire`); ``submit_device`` works on HBM-resident buffers
(torch uint8 tensors or raw device pointers) without any host copy.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib
from .wire import LOG_REC, MSG_DTYPE, Workload


def _ptr(x):
    """Device pointer of a torch tensor / int."""
    if isinstance(x, int):
        return x
    return x.data_ptr()


def _hptr(x):
    """Host pointer of a numpy array / pinned torch tensor / int."""
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()


class Pinned:
    """A page-locked host buffer from dint_alloc_pinned, viewed as a numpy uint8 array."""

    def __init__(self, nbytes: int):
        self._L = _lib.load()
        p = C.c_void_p()
        _lib.check(self._L.dint_alloc_pinned(nbytes, C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes
        self.array = np.frombuffer((C.c_uint8 * nbytes).from_address(self.ptr), np.uint8)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._L.dint_free_pinned(self.ptr)
            self.ptr = None

    def __del__(self):
        # at interpreter exit the HIP runtime may already be gone (module teardown order): the process is ending anyway
        if not sys.is_finalizing():
            self.close()


class Engine:
    def __init__(self, workload: Workload, *, n_slots: int = 0, n_rows: int = 0, log_entries: int = 0,
                 device: int = -1, shard_index: int = 0, shard_count: int = 1, flags: int = 0, max_pass: int = 0,
                 pool_entries: int = 0):
        self._L = _lib.load()
        self.workload = Workload(workload)
        self.msg_dtype = MSG_DTYPE[self.workload]
        self.msg_size = self.msg_dtype.itemsize
        cfg = _lib.Config(abi_version=_lib.ABI_VERSION, workload=int(workload), device=device, flags=flags, n_slots=n_slots,
                          n_rows=n_rows, log_entries=log_entries, shard_index=shard_index, shard_count=shard_count,
                          max_pass=max_pass, pool_entries=pool_entries)
        h = C.c_void_p()
        _lib.check(self._L.dint_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.shard_index, self.shard_count = shard_index, max(1, shard_count)
        self.val_size = 8 if self.workload == Workload.SMALLBANK else 40
        self.pass_max = int(self._L.dint_max_pass(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.dint_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter exit the HIP runtime may already be gone (module teardown order): the process is ending anyway
        if not sys.is_finalizing():
            self.close()

    # ---- hot path ---------------------------------------------------------
    def submit(self, reqs: np.ndarray) -> np.ndarray:
        """Process a batch held in host memory; returns the reply array (same dtype)."""
        reqs = np.ascontiguousarray(reqs)
        assert reqs.dtype.itemsize in (1, self.msg_size)
        n = reqs.nbytes // self.msg_size
        out = np.empty_like(reqs)
        _lib.check(self._L.dint_submit(self._h, reqs.ctypes.data, n, out.ctypes.data))
        return out

    def submit_device(self, d_reqs, n: int, d_replies=None, stream: int = 0, ahead=None) -> None:
        """Enqueue a batch that already lives in HBM (asynchronous).  `ahead` = (d_reqs, n, d_replies) of the engine's NEXT
        submit_device call, complete in HBM already (dint_submit_device_ahead: its partition stage runs beside this batch's
        hot keys; that call must follow, with exactly these buffers)."""
        d_replies = d_reqs if d_replies is None else d_replies
        if ahead is None or not ahead[1]:
            _lib.check(self._L.dint_submit_device(self._h, _ptr(d_reqs), n, _ptr(d_replies), stream))
        else:
            nq, nn, nr = ahead
            _lib.check(self._L.dint_submit_device_ahead(self._h, _ptr(d_reqs), n, _ptr(d_replies), _ptr(nq), nn,
                                                        _ptr(nq if nr is None else nr), stream))

    def submit_async(self, reqs, n: int, replies) -> int:
        """Pipelined host submit on raw host pointers (page-locked for real overlap); returns the ticket."""
        t = C.c_uint64()
        _lib.check(self._L.dint_submit_async(self._h, _hptr(reqs), n, _hptr(replies), C.byref(t)))
        return t.value

    def wait(self, ticket: int):
        _lib.check(self._L.dint_wait(self._h, ticket))

    def sync(self):
        _lib.check(self._L.dint_sync(self._h))

    # ---- streams / multi-GPU routing (dint_amd.sharded.Router) ------------------------------------------
    @property
    def stream(self) -> int:
        return self._L.dint_engine_stream(self._h) or 0

    def stream_wait(self, other_stream: int):
        _lib.check(self._L.dint_stream_wait(self._h, other_stream))

    def stream_signal(self, other_stream: int):
        _lib.check(self._L.dint_stream_signal(self._h, other_stream))

    def route_pack(self, d_reqs, n: int, d_send, seg_cap: int, seg_stride: int, d_cnt, cnt_stride: int, d_slot,
                   stream: int = 0):
        _lib.check(self._L.dint_route_pack(self._h, _ptr(d_reqs), n, _ptr(d_send), seg_cap, seg_stride, _ptr(d_cnt),
                                           cnt_stride, _ptr(d_slot), stream))

    def route_unpack(self, d_back, seg_cap: int, seg_stride: int, d_slot, d_reqs, n: int, d_replies, stream: int = 0):
        _lib.check(self._L.dint_route_unpack(self._h, _ptr(d_back), seg_cap, seg_stride, _ptr(d_slot), _ptr(d_reqs), n,
                                             _ptr(d_replies), stream))

    def submit_segments(self, d_base, n_seg: int, seg_cap: int, seg_stride: int, d_cnt, cnt_stride: int, stream: int = 0):
        _lib.check(self._L.dint_submit_segments(self._h, _ptr(d_base), n_seg, seg_cap, seg_stride, _ptr(d_cnt),
                                                cnt_stride, stream))

    def home_shard(self, d_reqs, n: int, d_home, stream: int = 0) -> None:
        _lib.check(self._L.dint_home_shard(self._h, _ptr(d_reqs), n, _ptr(d_home), stream))

    # ---- population / state ---------------------------------------------------
    def populate(self, populate_n: int):
        _lib.check(self._L.dint_populate(self._h, populate_n))

    def load_rows(self, table: int, keys, vers, vals):
        keys = np.ascontiguousarray(keys, "<u8")
        vals = np.ascontiguousarray(vals, "u1").reshape(len(keys), self.val_size)
        vp = None
        if vers is not None:
            vers = np.ascontiguousarray(vers, "<u4")
            vp = vers.ctypes.data
        _lib.check(self._L.dint_load_rows(self._h, table, keys.ctypes.data, vp, vals.ctypes.data, len(keys)))

    def hash_size(self, table: int) -> int:
        return _lib.check(self._L.dint_hash_size(self._h, table))

    def dump_rows(self, table: int):
        n = _lib.check(self._L.dint_dump_rows(self._h, table, None, None, None, 0))
        keys = np.zeros(n, "<u8"); vers = np.zeros(n, "<u4"); vals = np.zeros((n, self.val_size), "u1")
        got = _lib.check(self._L.dint_dump_rows(self._h, table, keys.ctypes.data, vers.ctypes.data, vals.ctypes.data, n))
        assert got == n
        return keys, vers, vals

    def read_locks(self, table: int = 0):
        n = _lib.check(self._L.dint_read_locks(self._h, table, None, None, 0))
        a = np.zeros(n, "<u4"); b = np.zeros(n, "<u4")
        _lib.check(self._L.dint_read_locks(self._h, table, a.ctypes.data, b.ctypes.data, n))
        return a, b

    def read_log(self, cap: int):
        rec = np.zeros(cap, LOG_REC)
        tail = _lib.check(self._L.dint_read_log(self._h, rec.ctypes.data, cap))
        return rec, tail

    def log_drain(self, cap: int = 1 << 20):
        """records appended since the previous drain (oldest first) and how many were lost to ring overwrite"""
        rec = np.zeros(cap, LOG_REC)
        lost = C.c_uint64()
        n = _lib.check(self._L.dint_log_drain(self._h, rec.ctypes.data, cap, C.byref(lost)))
        return rec[:n], lost.value

    def stats(self) -> dict:
        s = _lib.Stats()
        _lib.check(self._L.dint_get_stats(self._h, C.byref(s)))
        d = {k: getattr(s, k) for k, _ in s._fields_ if k != "reserved"}
        d["late_items"] = list(s.reserved)  # diagnostic: late work items by kind (a sub as listed / a solo item / pieces)
        return d

    def reset(self): _lib.check(self._L.dint_reset(self._h))
    def snapshot(self): _lib.check(self._L.dint_snapshot(self._h))
    def restore(self): _lib.check(self._L.dint_restore(self._h))

    # ---- measurement -----------------------------------------------------------
    def timing_enable(self, on: bool = True):
        _lib.check(self._L.dint_timing_enable(self._h, int(on)))

    def kv_trace(self, workgroups: bool = False):
        """DINT_KV_TRACE=1: [32768, 16] u64 per-bin timeline of the resolve launches since the last read; with
        workgroups=True also [8192, 16] per workgroup: {first wave in, last wave out, phase stamps of its first big bin}
        (10 ns ticks)."""
        out = np.zeros(32768 * 16 + 16 * 8192, "<u8")
        _lib.check(self._L.dint_kv_trace_read(self._h, out.ctypes.data, out.size))
        if self.workload in (Workload.STORE, Workload.TATP, Workload.SMALLBANK):
            # kv engines (r04): 32 words per resolve workgroup -- [0] in, [1] descriptors / count / first records there,
            # [2] records counted per sub, [3] laid out, [4] placed, [5] first chunk sorted, [6..10] that chunk's kv_chunk
            # stamps (run / segment masks, rows located, replies, write-backs, rounds), [11] chunks done, [12] big subs
            # done, [13] coarse bin, [14] records, [15] records in big subs, [16] chunks
            wg = out[:2048 * 32].reshape(2048, 32)
            # ... and 32 words per k_kv_big workgroup: {in, out, records, sub} of the first big sub it resolved, then the
            # phase stamps of one of its stretches ([4] in .. [13] out, [16] .. [22] its dominant-key path: kv_big_bin)
            return (wg, out[2048 * 32:2048 * 32 + 4 * 512 * 32].reshape(2048, 32)) if workgroups else wg
        bins = out[:32768 * 16].reshape(32768, 16)
        return (bins, out[32768 * 16:].reshape(8192, 16)) if workgroups else bins

    def timing_read(self) -> dict:
        names = (C.c_char_p * 8)(); us = (C.c_double * 8)(); cnt = (C.c_uint64 * 8)()
        k = _lib.check(self._L.dint_timing_read(self._h, names, us, cnt, 8))
        return {names[i].decode(): {"avg_us": us[i], "launches": cnt[i]} for i in range(k)}


def submit_segments_multi(engines, d_bases, n_seg: int, seg_caps, seg_stride: int, d_cnts, cnt_stride: int, stream: int = 0, ahead=None):
    """dint_submit_segments_multi: engine k answers its n_seg segments at d_bases[k] in place; the engines' kernels run
    side by side in ONE set of launches on ONE stream (no fork / join across the engines' streams).
    `ahead` = (d_bases, d_cnts) of the NEXT call (same engines, same geometry; dint_submit_segments_multi_ahead)."""
    items = (_lib.SegmentsItem * len(engines))()
    for k, e in enumerate(engines):
        items[k] = _lib.SegmentsItem(e._h, _ptr(d_bases[k]), n_seg, seg_caps[k], seg_stride, _ptr(d_cnts[k]), cnt_stride)
    if ahead is None:
        _lib.check(engines[0]._L.dint_submit_segments_multi(items, len(engines), stream))
        return
    nxt = (_lib.SegmentsItem * len(engines))()
    for k, e in enumerate(engines):
        nxt[k] = _lib.SegmentsItem(e._h, _ptr(ahead[0][k]), n_seg, seg_caps[k], seg_stride, _ptr(ahead[1][k]), cnt_stride)
    _lib.check(engines[0]._L.dint_submit_segments_multi_ahead(items, len(engines), nxt, stream))


def route_pack_multi(engines, d_reqs, counts, d_slots, seg_caps, seg_stride: int, d_cnts, cnt_stride: int, d_slot,
                     stream: int = 0, d_n=None) -> None:
    """dint_route_pack_multi: batch k (engine k's hash / modulus) into its slot d_slots[k] of every peer chunk, one set
    of kernel launches for all of them.  d_n[k]: device pointer to batch k's live request count (counts[k] is then
    its upper bound)."""
    items = (_lib.RouteItem * len(engines))()
    for k, e in enumerate(engines):
        items[k] = _lib.RouteItem(e._h, _ptr(d_reqs[k]), counts[k], seg_caps[k], _ptr(d_slots[k]), _ptr(d_cnts[k]),
                                  _ptr(d_slot[k]), None, d_n[k] if d_n else None)
    _lib.check(engines[0]._L.dint_route_pack_multi(items, len(engines), seg_stride, cnt_stride, stream))


def route_unpack_multi(engines, d_backs, seg_caps, seg_stride: int, d_slot, d_reqs, counts, d_replies, stream: int = 0,
                       d_n=None) -> None:
    items = (_lib.RouteItem * len(engines))()
    for k, e in enumerate(engines):
        items[k] = _lib.RouteItem(e._h, _ptr(d_reqs[k]), counts[k], seg_caps[k], _ptr(d_backs[k]), None, _ptr(d_slot[k]),
                                  _ptr(d_replies[k]), d_n[k] if d_n else None)
    _lib.check(engines[0]._L.dint_route_unpack_multi(items, len(engines), seg_stride, stream))


def bench_rand64(bytes_: int, n_access: int, write_back: bool = False, device: int = -1):
    """Random 64-byte gather microbenchmark (roofline denominator); returns (accesses/s, seconds)."""
    L = _lib.load()
    aps = C.c_double(); sec = C.c_double()
    _lib.check(L.dint_bench_rand64(device, bytes_, n_access, int(write_back), C.byref(aps), C.byref(sec)))
    return aps.value, sec.value


BENCH_MODES = {"gather": 0, "rmw": 1, "scatter": 2, "atomic": 3, "atomic_ret": 4, "stream_rd": 5, "stream_wr": 6}


def bench_access(bytes_: int, n_access: int, mode: str, width: int = 64, blocks_per_cu: int = 8, device: int = -1):
    """One access pattern of csrc/k_bench.hip over `bytes_` of HBM; returns (accesses/s, seconds) -- for the two
    streaming modes 16-byte vectors/s."""
    L = _lib.load()
    aps = C.c_double(); sec = C.c_double()
    _lib.check(L.dint_bench_access(device, bytes_, n_access, BENCH_MODES[mode], width, blocks_per_cu,
                                   C.byref(aps), C.byref(sec)))
    return aps.value, sec.value


def refuse(workload, reqs: np.ndarray) -> np.ndarray:
    """the back-pressure replies of the eBPF-flavour servers (REJECT_* / RETRY) for a batch that is not taken"""
    L = _lib.load()
    reqs = np.ascontiguousarray(reqs)
    out = np.empty_like(reqs)
    _lib.check(L.dint_refuse(int(workload), reqs.ctypes.data, len(reqs), out.ctypes.data))
    return out
