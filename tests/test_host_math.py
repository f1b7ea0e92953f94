"""Integer arithmetic the device code relies on, checked on the host (tests/native/host_math.cc): the indexed Zipf
lookup against the plain binary search (the host and device drivers must draw the same keys), and the magic-multiply
division behind the bin cut of the kv passes (bin = group % P for ANY P)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "host_math.cc")
LIB = os.path.join(HERE, "native", "libhost_math.so")


@pytest.fixture(scope="module")
def hm():
    hdr = os.path.join(HERE, "..", "dint_amd", "csrc", "zipf_table.h")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", LIB, SRC])
    L = C.CDLL(LIB)
    L.zipf_mismatches.restype, L.zipf_mismatches.argtypes = C.c_uint64, [C.c_uint64, C.c_double, C.c_void_p, C.c_uint64]
    L.cut_mismatches.restype, L.cut_mismatches.argtypes = C.c_uint64, [C.c_uint32, C.c_void_p, C.c_uint64]
    return L


@pytest.mark.parametrize("n,theta", [(1, 0.8), (2, 0.5), (4800, 0.8), (1_000_000, 0.8), (1_000_000, 0.99 - 1e-9), (3_000_001, 0.2)])
def test_indexed_zipf_lookup_equals_plain_binary_search(hm, n, theta):
    rng = np.random.default_rng(n)
    xs = rng.integers(0, 1 << 32, 400_000, dtype=np.uint64).astype(np.uint32)
    # the edges of every index bucket and of the 32-bit range
    edges = (np.arange(0, 65537, dtype=np.uint64) << np.uint64(16))  # ZIPF_IDX_BITS = 16
    edges = np.clip(np.concatenate([edges, edges + 1, edges - 1]), 0, (1 << 32) - 1).astype(np.uint32)
    xs = np.ascontiguousarray(np.concatenate([xs, edges]))
    assert hm.zipf_mismatches(n, theta, xs.ctypes.data, len(xs)) == 0


@pytest.mark.parametrize("P", [1, 2, 3, 7, 31, 32, 33, 4096, 7499, 8192, 10_007, 32_767, 32_768])
def test_bin_cut_division_is_exact(hm, P):
    rng = np.random.default_rng(P)
    gks = rng.integers(0, 1 << 32, 500_000, dtype=np.uint64).astype(np.uint32)
    edge = np.array([0, 1, P - 1, P, P + 1, 2 * P - 1, (1 << 32) - 1, (1 << 32) - P, ((1 << 32) // P) * P - 1, ((1 << 32) // P) * P % (1 << 32)],
                    dtype=np.uint64).astype(np.uint32)
    gks = np.ascontiguousarray(np.concatenate([gks, edge]))
    assert hm.cut_mismatches(P, gks.ctypes.data, len(gks)) == 0
