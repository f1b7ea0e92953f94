"""ctypes binding of the CPU parity oracle (oracle/libdint_oracle.so) and a runner
for the unmodified-reference replay binaries under oracle/_ref/.

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  Nothing under dint_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdint_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")


def build(force: bool = False) -> None:
    """Compile the C restatement (and, where /root/reference exists, the _ref binaries)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
        os.path.join(HERE, "dint_oracle.c")
    ):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/lock_fasst/udp"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", "ref_ebpf", "ref_client"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
        sig = {
            "orc_fasthash64": (u64, [vp, u64, u64]),
            "orc_fasst_create": (vp, [u32]), "orc_fasst_destroy": (None, [vp]),
            "orc_fasst_replay": (u64, [vp, vp, sz]),
            "orc_fasst_locks": (vp, [vp]), "orc_fasst_vers": (vp, [vp]),
            "orc_2pl_create": (vp, [u32]), "orc_2pl_destroy": (None, [vp]),
            "orc_2pl_replay": (u64, [vp, vp, sz]),
            "orc_2pl_num_ex": (vp, [vp]), "orc_2pl_num_sh": (vp, [vp]),
            "orc_log_create": (vp, [u32]), "orc_log_destroy": (None, [vp]),
            "orc_log_replay": (u64, [vp, vp, sz]),
            "orc_log_ring": (vp, [vp]), "orc_log_tail": (u32, [vp]),
            "orc_kvs_create": (vp, [u32, u32]), "orc_kvs_destroy": (None, [vp]),
            "orc_kvs_get": (C.c_int, [vp, u64, vp, vp]), "orc_kvs_set": (C.c_int, [vp, u64, vp]),
            "orc_kvs_insert": (None, [vp, u64, vp]), "orc_kvs_delete": (C.c_int, [vp, u64]),
            "orc_kvs_count": (u64, [vp]),
            "orc_kvs_dump": (u64, [vp, vp, vp, vp, u64]),
            "orc_kvs_load": (None, [vp, vp, vp, vp, u64]),
            "orc_store_create": (vp, [u32, u32]), "orc_store_destroy": (None, [vp]),
            "orc_store_replay": (u64, [vp, vp, sz]), "orc_store_table": (vp, [vp]),
            "orc_tatp_create": (vp, [u32, u32, u32]), "orc_tatp_destroy": (None, [vp]),
            "orc_tatp_replay": (u64, [vp, vp, sz]),
            "orc_tatp_table": (vp, [vp, C.c_int]), "orc_tatp_hash_size": (u32, [vp, C.c_int]),
            "orc_tatp_locks": (vp, [vp, C.c_int]), "orc_tatp_same_key_mode": (None, [vp]),
            "orc_tatp_log_ring": (vp, [vp]), "orc_tatp_log_tail": (u32, [vp]),
            "orc_sb_create": (vp, [u32, u32, u32]), "orc_sb_destroy": (None, [vp]),
            "orc_sb_replay": (u64, [vp, vp, sz]),
            "orc_sb_table": (vp, [vp, C.c_int]), "orc_sb_hash_size": (u32, [vp, C.c_int]),
            "orc_sb_num_ex": (vp, [vp, C.c_int]), "orc_sb_num_sh": (vp, [vp, C.c_int]),
            "orc_sb_log_ring": (vp, [vp]), "orc_sb_log_tail": (u32, [vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def fasthash64(data: bytes, seed: int = 0xDEADBEEF) -> int:
    buf = C.create_string_buffer(data, len(data))
    return lib().orc_fasthash64(C.cast(buf, C.c_void_p), len(data), seed)


def _view(ptr: int, n: int, dtype) -> np.ndarray:
    dt = np.dtype(dtype)
    buf = (C.c_uint8 * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n)


def _bytes_inplace(msgs: np.ndarray, itemsize: int):
    assert msgs.flags["C_CONTIGUOUS"] and msgs.dtype.itemsize in (1, itemsize)
    n = msgs.nbytes // itemsize
    return msgs.ctypes.data, n


def _dump_kvs(kvs_ptr: int, val_size: int):
    L = lib()
    n = L.orc_kvs_count(kvs_ptr)
    keys = np.zeros(n, "<u8"); vers = np.zeros(n, "<u4"); vals = np.zeros((n, val_size), "u1")
    got = L.orc_kvs_dump(kvs_ptr, keys.ctypes.data, vers.ctypes.data, vals.ctypes.data, n)
    assert got == n
    return keys, vers, vals


class KvsOracle:
    """One chained 4-way table (store/udp/kvs.h) on its own -- used to pin the engine's HBM layout."""

    def __init__(self, hash_size: int, val_size: int):
        self.vs = val_size
        self.h = lib().orc_kvs_create(hash_size, val_size)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kvs_destroy(self.h)
            self.h = None

    def get(self, key: int):
        val = np.zeros(self.vs, "u1"); ver = C.c_uint32()
        rc = lib().orc_kvs_get(self.h, key, val.ctypes.data, C.addressof(ver))
        return (None, None) if rc else (val, ver.value)

    def set(self, key: int, val: np.ndarray) -> int:
        return lib().orc_kvs_set(self.h, key, np.ascontiguousarray(val, "u1").ctypes.data)

    def insert(self, key: int, val: np.ndarray) -> None:
        lib().orc_kvs_insert(self.h, key, np.ascontiguousarray(val, "u1").ctypes.data)

    def delete(self, key: int) -> int:
        return lib().orc_kvs_delete(self.h, key)

    def dump(self):
        return _dump_kvs(self.h, self.vs)


class _Base:
    _destroy = None
    ITEM = 0

    def __del__(self):
        if getattr(self, "h", None):
            getattr(lib(), self._destroy)(self.h)
            self.h = None

    def replay(self, msgs: np.ndarray) -> np.ndarray:
        """Serially process a batch (copy); returns the reply array; self.errors accumulates."""
        out = np.ascontiguousarray(msgs).copy()
        ptr, n = _bytes_inplace(out, self.ITEM)
        self.errors = getattr(self, "errors", 0) + getattr(lib(), self._replay)(self.h, ptr, n)
        return out


class FasstOracle(_Base):
    _destroy, _replay, ITEM = "orc_fasst_destroy", "orc_fasst_replay", 9

    def __init__(self, nslots: int):
        self.n = nslots
        self.h = lib().orc_fasst_create(nslots)

    @property
    def locks(self): return _view(lib().orc_fasst_locks(self.h), self.n, "<u4")

    @property
    def vers(self): return _view(lib().orc_fasst_vers(self.h), self.n, "<u4")


class TplOracle(_Base):
    _destroy, _replay, ITEM = "orc_2pl_destroy", "orc_2pl_replay", 6

    def __init__(self, nslots: int):
        self.n = nslots
        self.h = lib().orc_2pl_create(nslots)

    @property
    def num_ex(self): return _view(lib().orc_2pl_num_ex(self.h), self.n, "<u4")

    @property
    def num_sh(self): return _view(lib().orc_2pl_num_sh(self.h), self.n, "<u4")


class LogOracle(_Base):
    _destroy, _replay, ITEM = "orc_log_destroy", "orc_log_replay", 53

    def __init__(self, ring_entries: int = 1_000_000):
        self.cap = ring_entries
        self.h = lib().orc_log_create(ring_entries)

    @property
    def ring(self): return _view(lib().orc_log_ring(self.h), self.cap * 64, "u1").reshape(self.cap, 64)

    @property
    def tail(self): return lib().orc_log_tail(self.h)


class StoreOracle(_Base):
    _destroy, _replay, ITEM = "orc_store_destroy", "orc_store_replay", 53

    def __init__(self, hash_size: int, populate_n: int = 0):
        """hash_size buckets; rows of the first `populate_n` subscribers (store/udp/tatp.h:44-66)."""
        self.h = lib().orc_store_create(hash_size, populate_n)

    def dump(self):
        return _dump_kvs(lib().orc_store_table(self.h), 40)

    def load(self, keys, vers, vals):
        keys = np.ascontiguousarray(keys, "<u8"); vers = np.ascontiguousarray(vers, "<u4")
        vals = np.ascontiguousarray(vals, "u1")
        lib().orc_kvs_load(lib().orc_store_table(self.h), keys.ctypes.data, vers.ctypes.data,
                           vals.ctypes.data, len(keys))


class TatpOracle(_Base):
    _destroy, _replay, ITEM = "orc_tatp_destroy", "orc_tatp_replay", 55

    def __init__(self, n_sub: int, log_entries: int = 1_000_000, populate_n: int | None = None):
        """`n_sub` sizes the tables; rows of the first `populate_n` (default all) subscribers."""
        self.cap = log_entries
        self.h = lib().orc_tatp_create(n_sub, log_entries, n_sub if populate_n is None else populate_n)

    def hash_size(self, t: int) -> int: return lib().orc_tatp_hash_size(self.h, t)

    def same_key_mode(self): lib().orc_tatp_same_key_mode(self.h)  # tatp/ebpf/lock_kern.c flavour

    def dump(self, t: int): return _dump_kvs(lib().orc_tatp_table(self.h, t), 40)

    def load(self, t: int, keys, vers, vals):
        keys = np.ascontiguousarray(keys, "<u8"); vers = np.ascontiguousarray(vers, "<u4")
        vals = np.ascontiguousarray(vals, "u1")
        lib().orc_kvs_load(lib().orc_tatp_table(self.h, t), keys.ctypes.data, vers.ctypes.data,
                           vals.ctypes.data, len(keys))

    def locks(self, t: int): return _view(lib().orc_tatp_locks(self.h, t), 4 * self.hash_size(t), "u1")

    @property
    def ring(self): return _view(lib().orc_tatp_log_ring(self.h), self.cap * 64, "u1").reshape(self.cap, 64)

    @property
    def tail(self): return lib().orc_tatp_log_tail(self.h)


class SmallbankOracle(_Base):
    _destroy, _replay, ITEM = "orc_sb_destroy", "orc_sb_replay", 23

    def __init__(self, n_acct: int, log_entries: int = 1_000_000, populate_n: int | None = None):
        self.cap = log_entries
        self.h = lib().orc_sb_create(n_acct, log_entries, n_acct if populate_n is None else populate_n)

    def hash_size(self, t: int) -> int: return lib().orc_sb_hash_size(self.h, t)

    def dump(self, t: int): return _dump_kvs(lib().orc_sb_table(self.h, t), 8)

    def load(self, t: int, keys, vers, vals):
        keys = np.ascontiguousarray(keys, "<u8"); vers = np.ascontiguousarray(vers, "<u4")
        vals = np.ascontiguousarray(vals, "u1")
        lib().orc_kvs_load(lib().orc_sb_table(self.h, t), keys.ctypes.data, vers.ctypes.data,
                           vals.ctypes.data, len(keys))

    def num_ex(self, t: int): return _view(lib().orc_sb_num_ex(self.h, t), 4 * self.hash_size(t), "<u4")

    def num_sh(self, t: int): return _view(lib().orc_sb_num_sh(self.h, t), 4 * self.hash_size(t), "<u4")

    @property
    def ring(self): return _view(lib().orc_sb_log_ring(self.h), self.cap * 64, "u1").reshape(self.cap, 64)

    @property
    def tail(self): return lib().orc_sb_log_tail(self.h)


# --------------------------------------------------------------------------- #
# unmodified reference, replayed through the socket-interposing harness
REF_BIN = {
    "lock_fasst": "ref_lock_fasst", "lock_2pl": "ref_lock_2pl", "log_server": "ref_log_server",
    "store": "ref_store", "tatp": "ref_tatp", "smallbank": "ref_smallbank",
}


def ref_available(workload: str) -> bool:
    return os.access(os.path.join(REF_DIR, REF_BIN[workload]), os.X_OK)


def ref_replay(workload: str, msgs: np.ndarray, dump: bool = False, timeout: float = 3600):
    """Run the unmodified reference server over `msgs` (one thread, serial).

    Returns (replies ndarray of msgs.dtype, stats dict[, dump bytes]).
    """
    exe = os.path.join(REF_DIR, REF_BIN[workload])
    msgs = np.ascontiguousarray(msgs)
    with tempfile.TemporaryDirectory(prefix="dint_ref_") as td:
        tp, rp, dp = (os.path.join(td, x) for x in ("trace.bin", "replies.bin", "dump.bin"))
        msgs.tofile(tp)
        cmd = [exe, tp, rp] + ([dp] if dump else [])
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        if res.returncode != 0:
            raise RuntimeError(f"{exe} failed rc={res.returncode}: {res.stderr[-2000:]}")
        stats = json.loads(res.stdout.strip().splitlines()[-1])
        replies = np.fromfile(rp, dtype=msgs.dtype)
        if dump:
            with open(dp, "rb") as f:
                return replies, stats, f.read()
        return replies, stats


#: the eBPF flavour (DINT proper): unmodified <wl>/ebpf/*_kern.c + *_user.c under ref_harness/ebpf's emulator
EBPF_BIN = {
    "lock_fasst": "ref_ebpf_lock_fasst", "lock_2pl": "ref_ebpf_lock_2pl", "log_server": "ref_ebpf_log_server",
    "tatp_lock": "ref_ebpf_tatp_lock", "tatp": "ref_ebpf_tatp", "smallbank": "ref_ebpf_smallbank",
    "store": "ref_ebpf_store",
}


def ebpf_available(workload: str) -> bool:
    return os.access(os.path.join(REF_DIR, EBPF_BIN[workload]), os.X_OK)


def ebpf_replay(workload: str, msgs: np.ndarray, timeout: float = 3600, hold: int = 0, hold_every: int = 1):
    """Run the unmodified reference eBPF server of `workload` (XDP program -> user-space fallback -> TC program,
    oracle/ref_harness/ebpf) over `msgs`, one request at a time.  Returns (replies, stats); a request the server
    never answers comes back unchanged and is counted in stats["unanswered"].
    hold: during every `hold_every`-th request the emulator holds the spin lock of every cache entry (bit 0) / lock unit
    (bit 1) the XDP program looks up, as a concurrent packet would -- what the program answers then are the back-pressure
    replies a serial replay never sees (REJECT_READ / REJECT_COMMIT / REJECT_SET / RETRY ...)."""
    exe = os.path.join(REF_DIR, EBPF_BIN[workload])
    msgs = np.ascontiguousarray(msgs)
    with tempfile.TemporaryDirectory(prefix="dint_ebpf_") as td:
        tp, rp = os.path.join(td, "trace.bin"), os.path.join(td, "replies.bin")
        msgs.tofile(tp)
        env = dict(os.environ)
        if hold:
            env.update(EMU_HOLD=str(hold), EMU_HOLD_EVERY=str(hold_every))
        res = subprocess.run([exe, tp, rp], capture_output=True, text=True, timeout=timeout, env=env)
        if res.returncode != 0:
            raise RuntimeError(f"{exe} failed rc={res.returncode}: {res.stderr[-2000:]}")
        return np.fromfile(rp, dtype=msgs.dtype), json.loads(res.stdout.strip().splitlines()[-1])


class RefServer:
    """The unmodified reference server started ahead of its trace (REF_TRACE_WAIT, ref_harness/harness_common.h):
    tatp / smallbank populate for minutes, so a caller starts the server first, does its own work, and hands the
    trace over when it has one.  `replay` blocks until the server has populated, replayed and exited."""

    def __init__(self, workload: str):
        self.workload = workload
        self.td = tempfile.TemporaryDirectory(prefix="dint_ref_")
        self.tp, self.rp = (os.path.join(self.td.name, x) for x in ("trace.bin", "replies.bin"))
        env = dict(os.environ, REF_TRACE_WAIT="1")
        self.t0 = time.perf_counter()
        self.proc = subprocess.Popen([os.path.join(REF_DIR, REF_BIN[workload]), self.tp, self.rp],
                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
        self.populate_s = None

    def populated(self) -> bool:
        return os.path.exists(self.rp + ".ready")

    def wait_populated(self, timeout: float) -> bool:
        end = time.perf_counter() + timeout
        while not self.populated():
            if self.proc.poll() is not None or time.perf_counter() > end:
                return False
            time.sleep(0.05)
        if self.populate_s is None:
            self.populate_s = time.perf_counter() - self.t0
        return True

    def replay(self, msgs: np.ndarray, timeout: float = 600):
        msgs = np.ascontiguousarray(msgs)
        msgs.tofile(self.tp)
        open(self.tp + ".go", "w").close()
        out, _ = self.proc.communicate(timeout=timeout)
        if self.proc.returncode != 0:
            raise RuntimeError(f"reference {self.workload} server failed rc={self.proc.returncode}")
        stats = json.loads(out.strip().splitlines()[-1])
        return np.fromfile(self.rp, dtype=msgs.dtype), stats

    def close(self):
        if self.proc.poll() is None:
            self.proc.kill()  # the exact process this object started
            self.proc.wait()
        self.td.cleanup()


def loopback_available() -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, b)) for b in ("udp_lock_fasst_server", "udp_loop_client"))


def ref_loopback_fasst(msgs: np.ndarray, server_threads: int = 8, client_threads: int = 16, window: int = 32,
                       warmup_s: float = 1.0, measure_s: float = 5.0) -> dict:
    """the as-shipped reference `lock_fasst/udp/server <T>` (unmodified, its own main() and thread
    pinning, kernel UDP sockets on 127.0.0.1) driven by ref_harness/udp_loop_client.c replaying `msgs` closed-loop."""
    with tempfile.TemporaryDirectory(prefix="dint_ref_") as td:
        tp = os.path.join(td, "requests.bin")
        np.ascontiguousarray(msgs).tofile(tp)
        srv = subprocess.Popen([os.path.join(REF_DIR, "udp_lock_fasst_server"), str(server_threads)],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            time.sleep(0.5)
            if srv.poll() is not None:
                raise RuntimeError(f"reference udp server exited rc={srv.returncode} (port 20230 busy?)")
            res = subprocess.run([os.path.join(REF_DIR, "udp_loop_client"), tp, str(msgs.dtype.itemsize), "20230",
                                  str(client_threads), str(window), str(warmup_s), str(measure_s)],
                                 capture_output=True, text=True, timeout=warmup_s + measure_s + 60)
            if res.returncode != 0:
                raise RuntimeError(f"udp_loop_client failed rc={res.returncode}: {res.stderr[-500:]}")
            out = json.loads(res.stdout.strip().splitlines()[-1])
        finally:
            srv.kill()  # the exact process started above
            srv.wait()
    out["server_threads"] = server_threads
    return out


class TatpAsShipped:
    """BASELINE.md 3(2) for the headline workload: the reference's as-shipped deployment -- three `tatp/udp/server_shard <id>
    <T>` processes (unmodified server_shard.cc, its own main(), populate and thread pinning; only bind() is redirected:
    10.10.1.N -> 127.0.1.N, ref_harness/bind_lo.c).  Started ahead of its traffic (each populates 7M subscribers for tens of
    seconds); `run` drives each over loopback UDP with a closed-loop client replaying `streams[id - 1]`
    (udp_loop_client.c), all three at once, and returns the summed reply rate and the per-server lines."""

    def __init__(self, server_threads: int = 8):
        self.server_threads, self.t0, self.populate_s = server_threads, time.perf_counter(), None
        env = dict(os.environ, BIND_LO_MAP="1")
        self.procs = [subprocess.Popen([os.path.join(REF_DIR, "udp_tatp_server"), str(sid), str(server_threads)], env=env,
                                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for sid in (1, 2, 3)]

    def wait_populated(self, timeout: float = 240.0) -> bool:
        import select

        for p in self.procs:  # "finish initialization" = populated (server_shard.cc:294)
            while True:
                left = timeout - (time.perf_counter() - self.t0)
                if left <= 0 or not select.select([p.stdout], [], [], left)[0]:
                    return False
                line = p.stdout.readline()
                if not line:
                    raise RuntimeError(f"reference tatp server exited rc={p.poll()} (port 20230 on 127.0.1.N busy?)")
                if "finish initialization" in line:
                    break
        self.populate_s = round(time.perf_counter() - self.t0, 1)
        time.sleep(0.5)  # the worker threads bind their sockets
        return True

    def run(self, streams, client_threads: int = 16, window: int = 32, warmup_s: float = 1.0, measure_s: float = 4.0) -> dict:
        out = {"servers": []}
        with tempfile.TemporaryDirectory(prefix="dint_ref_") as td:
            for sid in (1, 2, 3):
                np.ascontiguousarray(streams[sid - 1]).tofile(os.path.join(td, f"req{sid}.bin"))
            cl = [subprocess.Popen([os.path.join(REF_DIR, "udp_loop_client"), os.path.join(td, f"req{sid}.bin"),
                                    str(streams[sid - 1].dtype.itemsize), "20230", str(client_threads), str(window), str(warmup_s),
                                    str(measure_s), f"127.0.1.{sid}"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                  for sid in (1, 2, 3)]
            for c in cl:
                so, se = c.communicate(timeout=warmup_s + measure_s + 60)
                if c.returncode != 0:
                    raise RuntimeError(f"udp_loop_client failed rc={c.returncode}: {se[-300:]}")
                out["servers"].append(json.loads(so.strip().splitlines()[-1]))
        if any(p.poll() is not None for p in self.procs):
            raise RuntimeError("a reference tatp server exited during the run (a request it panics on?)")
        out["ops_per_s"] = sum(x["ops_per_s"] for x in out["servers"])
        out["lost"] = sum(x["lost"] for x in out["servers"])
        out["server_threads"], out["populate_s"] = self.server_threads, self.populate_s
        return out

    def close(self):
        for p in self.procs:  # the exact processes started above
            if p.poll() is None:
                p.kill()
            p.wait()
        self.procs = []


def loopback_tatp_available() -> bool:
    return os.access(os.path.join(REF_DIR, "udp_tatp_server"), os.X_OK) and os.access(os.path.join(REF_DIR, "udp_loop_client"), os.X_OK)


# ---------------------------------------------------------------------------- populate-time garbage
_STORE_GRANT_READ = 3  # store/udp/net.h:22 kGrantRead
_TATP_GRANT_READ = 4  # tatp/udp/net.h:23 kGrantRead
#: value bytes the reference's populate code assigns (everything else is stack
#: garbage in the reference, zero in the oracle/engine): tatp/udp/tatp.h:283-412
TATP_ASSIGNED = {
    0: [i for i in range(40) if not 8 <= i <= 14],
    1: [0, 1, 2, 3, 4],
    2: [0],
    3: [0, 3],
    4: [0, 1],
}
STORE_ASSIGNED = [0, 1]  # store/udp/tatp.h:57-59


def mask_populate_garbage(workload: str, rep: np.ndarray) -> np.ndarray:
    """Zero the unassigned value bytes of GRANT_READ replies with ver == 0 (rows that
    may still hold the reference's populate-time stack garbage)."""
    rep = rep.copy()
    if workload == "store":
        sel = (rep["type"] == _STORE_GRANT_READ) & (rep["ver"] == 0)
        keep = np.zeros(40, bool)
        keep[STORE_ASSIGNED] = True
        v = rep["val"]
        v[np.ix_(sel, ~keep)] = 0
        rep["val"] = v
    elif workload == "tatp":
        v = rep["val"]
        for t, cols in TATP_ASSIGNED.items():
            sel = (rep["type"] == _TATP_GRANT_READ) & (rep["ver"] == 0) & (rep["table"] == t)
            keep = np.zeros(40, bool)
            keep[cols] = True
            v[np.ix_(sel, ~keep)] = 0
        rep["val"] = v
    return rep
