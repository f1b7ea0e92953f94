"""The HBM table layout (dint_amd/csrc/dint_kv_core.h: inline entry + overflow pool, chain head in
the inline header) compiled for the host and driven op by op against the oracle's chained kvs
(store/udp/kvs.h semantics).  Same source the HIP kernels run; catches chain-order, prepend,
free/unlink, pool-recycling and duplicate-key bugs without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "kv_core_host.cc")
LIB = os.path.join(HERE, "native", "libkv_core_host.so")


@pytest.fixture(scope="module")
def kvh():
    hdr = os.path.join(HERE, "..", "dint_amd", "csrc", "dint_kv_core.h")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", LIB, SRC])
    L = C.CDLL(LIB)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.kvh_create.restype, L.kvh_create.argtypes = vp, [u64, u32, u32]
    L.kvh_destroy.argtypes = [vp]
    L.kvh_get.argtypes = [vp, u64, u64, vp, vp]
    L.kvh_set.argtypes = [vp, u64, u64, vp]
    L.kvh_insert.argtypes = [vp, u64, u64, vp, u32]
    L.kvh_delete.argtypes = [vp, u64, u64]
    L.kvh_rotate.argtypes = [vp]
    L.kvh_pool_top.restype, L.kvh_pool_top.argtypes = u32, [vp]
    L.kvh_set_lock_bytes.argtypes = [vp, u64, u32]
    L.kvh_get_lock_bytes.restype, L.kvh_get_lock_bytes.argtypes = u32, [vp, u64]
    L.kvh_dump.restype, L.kvh_dump.argtypes = u64, [vp, vp, vp, vp, u64]
    return L


def bucket_of(key, nb):
    return orc.fasthash64(int(key).to_bytes(8, "little")) % nb


def dump(L, h, vs):
    n = L.kvh_dump(h, None, None, None, 0)
    keys = np.zeros(n, "<u8"); vers = np.zeros(n, "<u4"); vals = np.zeros((n, vs), "u1")
    assert L.kvh_dump(h, keys.ctypes.data, vers.ctypes.data, vals.ctypes.data, n) == n
    return keys, vers, vals


@pytest.mark.parametrize("vs,nb,nkeys,nops,seed,dups", [
    (40, 1, 30, 4000, 1, False), (40, 3, 60, 8000, 2, False), (8, 2, 40, 6000, 3, False),
    (40, 2, 24, 6000, 4, True), (8, 1, 12, 5000, 5, True), (40, 7, 200, 20000, 6, False),
])
def test_layout_matches_chained_kvs(kvh, vs, nb, nkeys, nops, seed, dups):
    """Random get/set/insert/delete; with dups=True inserts of existing keys create duplicate rows,
    which only an exact chain-order reproduction answers the same way as the reference."""
    L = kvh
    rng = np.random.default_rng(seed)
    h = L.kvh_create(nb, 4096 if dups else 64, vs)  # duplicate rows pile up: inserts outnumber deletes
    o = orc.KvsOracle(nb, vs)
    keys = rng.integers(1, 2**62, nkeys, dtype=np.uint64)
    live = {}
    for b in range(nb):
        L.kvh_set_lock_bytes(h, b, 0xA5000000 | b)
    try:
        for step in range(nops):
            k = int(keys[rng.integers(0, nkeys)])
            b = bucket_of(k, nb)
            op = rng.integers(0, 10)
            val = rng.integers(0, 256, vs, dtype=np.uint8)
            if op < 3:
                ov, over = o.get(k)
                gv = np.zeros(vs, "u1"); gver = C.c_uint32(0xDEAD)
                rc = L.kvh_get(h, b, k, gv.ctypes.data, C.addressof(gver))
                assert (rc == 1) == (ov is None)
                if ov is not None:
                    assert (gv == ov).all() and gver.value == over
            elif op < 5:
                assert L.kvh_set(h, b, k, val.ctypes.data) == o.set(k, val)
            elif op < 8:
                if not dups and live.get(k, 0) > 0:
                    continue
                o.insert(k, val)
                assert L.kvh_insert(h, b, k, val.ctypes.data, 0) == 0
                live[k] = live.get(k, 0) + 1
            else:
                rc = o.delete(k)
                assert L.kvh_delete(h, b, k) == rc
                if rc == 0:
                    live[k] -= 1
            if step % 97 == 0:
                L.kvh_rotate(h)  # a pass boundary: freed overflow entries become reusable
            if step % 500 == 0:
                a, bb = dump(L, h, vs), o.dump()
                assert all((x == y).all() for x, y in zip(a, bb)), step
        a, bb = dump(L, h, vs), o.dump()
        assert all((x == y).all() for x, y in zip(a, bb))
        for b in range(nb):  # lock bytes live in the inline header next to `head`: never clobbered
            assert L.kvh_get_lock_bytes(h, b) == (0xA5000000 | b)
        # recycling keeps the pool bounded: far fewer bump allocations than overflow inserts
        assert L.kvh_pool_top(h) <= (4096 if dups else 64)
    finally:
        L.kvh_destroy(h)


def test_pool_exhaustion_is_reported(kvh):
    L = kvh
    h = L.kvh_create(1, 2, 40)
    val = np.zeros(40, "u1")
    ok = [L.kvh_insert(h, 0, 100 + i, val.ctypes.data, 0) for i in range(16)]
    assert ok[:12] == [0] * 12 and ok[12:] == [1] * 4  # inline + 2 pool entries = 12 rows
    assert dump(L, h, 40)[0].size == 12
    L.kvh_destroy(h)
