"""CPU stand-in for one rank's sharded engine: the routing calls of dint_amd.engine.Engine (route_pack /
submit_segments / route_unpack) restated in numpy on HOST buffers, with a CPU oracle as the server.  Lets the
world-size-2 gloo tests run dint_amd.sharded.Router -- the exact orchestration the GPU path uses -- without a GPU.
Test infrastructure only."""
from __future__ import annotations

import ctypes as C

import numpy as np

from dint_amd import wire

M = np.uint64(0x880355F21E6D1965)


def _mix(h):
    h = h ^ (h >> np.uint64(23))
    h = h * np.uint64(0x2127599BF4325C37)
    return h ^ (h >> np.uint64(47))


def fasthash_lid(lid: np.ndarray) -> np.ndarray:  # fasthash64(&lid, 4, 0xdeadbeef), lock_fasst/udp/utils.h:23-53
    with np.errstate(over="ignore"):
        h = np.uint64(0xDEADBEEF) ^ (np.uint64(4) * M)
        h = (h ^ _mix(lid.astype(np.uint64))) * M
        return _mix(h)


def fasthash_key(key: np.ndarray) -> np.ndarray:  # fasthash64(&key, 8, 0xdeadbeef)
    with np.errstate(over="ignore"):
        h = np.uint64(0xDEADBEEF) ^ (np.uint64(8) * M)
        h = (h ^ _mix(key.astype(np.uint64))) * M
        return _mix(h)


def _buf(ptr: int, nbytes: int) -> np.ndarray:
    return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), np.uint8)


def _p(x) -> int:
    return x if isinstance(x, int) else x.data_ptr()


class ServerDouble:
    """`oracle`: an oracle.* object over the FULL key space (only home keys ever reach it); `home_of(msgs)` ->
    home rank per message."""

    def __init__(self, workload, oracle, world: int, rank: int, home_of, pass_max: int = 1 << 20):
        self.workload = wire.Workload(workload)
        self.dtype = wire.MSG_DTYPE[self.workload]
        self.msg_size = self.dtype.itemsize
        self.oracle, self.world, self.rank, self.home_of = oracle, world, rank, home_of
        self.pass_max = pass_max
        self.route_overflow = 0

    # ---- the calls Router makes ---------------------------------------------------------------------------------
    def route_pack(self, d_reqs, n, d_send, seg_cap, seg_stride, d_cnt, cnt_stride, d_slot, stream=0):
        msg = self.msg_size
        req = _buf(_p(d_reqs), n * msg).reshape(n, msg) if n else np.zeros((0, msg), np.uint8)
        slot = np.frombuffer((C.c_uint32 * max(n, 1)).from_address(_p(d_slot)), np.uint32)[:n]
        home = self.home_of(np.frombuffer(req.tobytes(), self.dtype)) if n else np.zeros(0, np.int64)
        for w in range(self.world):
            idx = np.nonzero(home == w)[0]  # ascending = stable
            keep = idx[:seg_cap]
            self.route_overflow += len(idx) - len(keep)
            _buf(_p(d_cnt) + w * cnt_stride, 4).view("<u4")[0] = len(keep)
            if len(keep):
                _buf(_p(d_send) + w * seg_stride, len(keep) * msg).reshape(-1, msg)[:] = req[keep]
            slot[keep] = w * seg_cap + np.arange(len(keep), dtype=np.uint32)
            slot[idx[seg_cap:]] = 0xFFFFFFFF

    def submit_segments(self, d_base, n_seg, seg_cap, seg_stride, d_cnt, cnt_stride, stream=0):
        msg = self.msg_size
        segs = []
        for k in range(n_seg):
            c = int(_buf(_p(d_cnt) + k * cnt_stride, 4).view("<u4")[0])
            assert c <= seg_cap
            segs.append(_buf(_p(d_base) + k * seg_stride, c * msg))
        allreq = np.frombuffer(b"".join(s.tobytes() for s in segs), self.dtype)
        out = np.frombuffer(self.oracle.replay(allreq).tobytes(), np.uint8) if len(allreq) else np.zeros(0, np.uint8)
        o = 0
        for s in segs:
            s[:] = out[o:o + len(s)]
            o += len(s)

    def route_unpack(self, d_back, seg_cap, seg_stride, d_slot, d_reqs, n, d_replies, stream=0):
        msg = self.msg_size
        if n == 0:
            return
        slot = np.frombuffer((C.c_uint32 * n).from_address(_p(d_slot)), np.uint32)
        req = _buf(_p(d_reqs), n * msg).reshape(n, msg)
        rep = _buf(_p(d_replies), n * msg).reshape(n, msg)
        over = np.nonzero(slot == 0xFFFFFFFF)[0]
        if len(over):  # not sent (slot full): the back-pressure reply, as k_route.hip rt_refuse (= dint_refuse, host code)
            from dint_amd.engine import refuse

            r = refuse(self.workload, np.frombuffer(req[over].tobytes(), self.dtype))
            rep[over] = np.frombuffer(r.tobytes(), np.uint8).reshape(-1, msg)
        for i in range(n):
            s = int(slot[i])
            if s == 0xFFFFFFFF:
                continue
            else:
                h, pos = divmod(s, seg_cap)
                rep[i] = _buf(_p(d_back) + h * seg_stride + pos * msg, msg)

    def stream_wait(self, other): pass
    def stream_signal(self, other): pass
    def sync(self): pass
    def stats(self): return {"route_overflow": self.route_overflow}


def lid_home(nslots: int, world: int):
    return lambda m: (fasthash_lid(m["lid"]) % np.uint64(nslots) % np.uint64(world)).astype(np.int64)


def kv_home(hash_sizes, world: int, self_rank: int):
    hs = np.array(list(hash_sizes) + [1] * 8, np.uint64)

    def f(m):
        t = m["table"].astype(np.int64) if "table" in m.dtype.names else np.zeros(len(m), np.int64)
        ok = t < len(hash_sizes)
        h = (fasthash_key(m["key"]) % hs[np.where(ok, t, 0)] % np.uint64(world)).astype(np.int64)
        return np.where(ok, h, self_rank)

    return f
