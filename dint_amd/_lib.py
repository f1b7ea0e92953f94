"""ctypes binding of the C ABI (include/dint_abi.h) exported by dint_amd/libdint.so.

There is deliberately no fallback: if the HIP extension is missing or fails to load,
importing the engine raises -- the product path never runs on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DINT_LIB_PATH") or os.path.join(HERE, "libdint.so")  # (DINT_LIB_PATH: same-box A/B runs of two builds, tools/)

ABI_VERSION = 4
#: dint_config.flags (include/dint_abi.h)
FLAG_KV_ROUNDS, FLAG_COPY_STREAMS, FLAG_LOCK_SAME_KEY, FLAG_KV_NO_HOT, FLAG_INPUTS_READY = 1, 2, 4, 8, 16
MICRO_BATCH = 65536

#: every symbol include/dint_abi.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "dint_engine_create", "dint_engine_destroy", "dint_msg_size", "dint_last_error", "dint_submit",
    "dint_submit_device", "dint_sync", "dint_load_rows", "dint_populate", "dint_hash_size", "dint_dump_rows",
    "dint_read_locks", "dint_read_log", "dint_get_stats", "dint_reset", "dint_snapshot", "dint_restore",
    "dint_home_shard", "dint_bench_rand64", "dint_timing_enable", "dint_timing_read", "dint_kv_trace_read",
    "dint_submit_async", "dint_wait", "dint_alloc_pinned", "dint_free_pinned", "dint_engine_stream", "dint_max_pass",
    "dint_stream_wait", "dint_stream_signal", "dint_route_pack", "dint_route_unpack", "dint_submit_segments",
    "dint_log_drain", "dint_refuse", "dint_route_pack_multi", "dint_route_unpack_multi", "dint_bench_access", "dint_selftest",
    "dint_submit_segments_multi", "dint_submit_device_ahead", "dint_submit_segments_multi_ahead",
]


class RouteItem(C.Structure):
    """dint_route_item (include/dint_abi.h)"""
    _fields_ = [("engine", C.c_void_p), ("d_reqs", C.c_void_p), ("n", C.c_uint32), ("seg_cap", C.c_uint32),
                ("d_slots", C.c_void_p), ("d_cnt", C.c_void_p), ("d_slot", C.c_void_p), ("d_replies", C.c_void_p),
                ("d_n", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("workload", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
        ("n_slots", C.c_uint64), ("n_rows", C.c_uint64), ("log_entries", C.c_uint32),
        ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("max_pass", C.c_uint32),
        ("pool_entries", C.c_uint32), ("reserved", C.c_uint32 * 3),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("batches", C.c_uint64), ("requests", C.c_uint64), ("bad_requests", C.c_uint64),
        ("missing_keys", C.c_uint64), ("foreign_requests", C.c_uint64), ("pool_exhausted", C.c_uint64),
        ("route_overflow", C.c_uint64), ("big_bin_requests", C.c_uint64), ("late_requests", C.c_uint64),
        ("reserved", C.c_uint64 * 3),
    ]


class DintError(RuntimeError):
    pass


_lib = None


class SegmentsItem(C.Structure):
    """dint_segments_item (include/dint_abi.h)"""
    _fields_ = [("engine", C.c_void_p), ("d_base", C.c_void_p), ("n_seg", C.c_uint32), ("seg_cap", C.c_uint32),
                ("seg_stride", C.c_uint64), ("d_cnt", C.c_void_p), ("cnt_stride", C.c_uint64)]


def load() -> C.CDLL:
    """Load libdint.so (built in-tree by dint_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # libdint.so is linked against libamdhip64; torch ships its own copy.  Loading torch first makes the two share
        # ONE HIP runtime (the same SONAME resolves to the copy already mapped) -- with two runtimes in the process the
        # one initialised second finds no device.  Only the ordering matters; no torch symbol is used here.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise DintError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -m dint_amd.build`); there is no CPU fallback"
        )
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64, C.c_int32
    sig = {
        "dint_engine_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
        "dint_engine_destroy": (None, [vp]),
        "dint_msg_size": (C.c_int, [u32]),
        "dint_last_error": (C.c_char_p, []),
        "dint_submit": (C.c_int, [vp, vp, u32, vp]),
        "dint_submit_device": (C.c_int, [vp, vp, u32, vp, vp]),
        "dint_submit_device_ahead": (C.c_int, [vp, vp, u32, vp, vp, u32, vp, vp]),
        "dint_sync": (C.c_int, [vp]),
        "dint_load_rows": (C.c_int, [vp, u32, vp, vp, vp, u64]),
        "dint_populate": (C.c_int, [vp, u64]),
        "dint_hash_size": (i64, [vp, u32]),
        "dint_dump_rows": (i64, [vp, u32, vp, vp, vp, u64]),
        "dint_read_locks": (i64, [vp, u32, vp, vp, u64]),
        "dint_read_log": (i64, [vp, vp, u64]),
        "dint_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
        "dint_reset": (C.c_int, [vp]),
        "dint_snapshot": (C.c_int, [vp]),
        "dint_restore": (C.c_int, [vp]),
        "dint_home_shard": (C.c_int, [vp, vp, u32, vp, vp]),
        "dint_bench_rand64": (C.c_int, [i32, u64, u64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "dint_selftest": (C.c_int, [i32]),
        "dint_bench_access": (C.c_int, [i32, u64, u64, u32, u32, u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "dint_kv_trace_read": (C.c_int, [vp, vp, u64]),
        "dint_timing_enable": (C.c_int, [vp, C.c_int]),
        "dint_timing_read": (C.c_int, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(u64), C.c_int]),
        "dint_submit_async": (C.c_int, [vp, vp, u32, vp, C.POINTER(u64)]),
        "dint_wait": (C.c_int, [vp, u64]),
        "dint_alloc_pinned": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
        "dint_free_pinned": (None, [vp]),
        "dint_engine_stream": (vp, [vp]),
        "dint_max_pass": (u32, [vp]),
        "dint_stream_wait": (C.c_int, [vp, vp]),
        "dint_stream_signal": (C.c_int, [vp, vp]),
        "dint_route_pack": (C.c_int, [vp, vp, u32, vp, u32, u64, vp, u64, vp, vp]),
        "dint_route_unpack": (C.c_int, [vp, vp, u32, u64, vp, vp, u32, vp, vp]),
        "dint_route_pack_multi": (C.c_int, [C.POINTER(RouteItem), u32, u64, u64, vp]),
        "dint_route_unpack_multi": (C.c_int, [C.POINTER(RouteItem), u32, u64, vp]),
        "dint_submit_segments": (C.c_int, [vp, vp, u32, u32, u64, vp, u64, vp]),
        "dint_submit_segments_multi": (C.c_int, [C.POINTER(SegmentsItem), u32, vp]),
        "dint_submit_segments_multi_ahead": (C.c_int, [C.POINTER(SegmentsItem), u32, C.POINTER(SegmentsItem), vp]),
        "dint_log_drain": (i64, [vp, vp, u64, C.POINTER(u64)]),
        "dint_refuse": (C.c_int, [u32, vp, u32, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)  # AttributeError here = the .so does not export the ABI
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise DintError(f"dint error {rc}: {load().dint_last_error().decode(errors='replace')}")
    return rc
