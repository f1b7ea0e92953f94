"""Pins for the CPU oracle: the known-answer vectors of SURVEY.md 8(c), all of which
were produced by executing reference code (hash/rand from tatp/udp/utils.h, the
reply streams from the unmodified lock_fasst/udp/server.cc and store/udp/server.cc)."""
import struct

import numpy as np
import pytest

from dint_amd import wire
from oracle import oracle as orc


def test_kat1_fasthash64_u32():
    # KAT-1: fasthash64(u32 lid, 4, 0xdeadbeef) and slot = hash % 36000000
    vec = {
        0: (0xF1D9C3BC57488240, 28482624),
        1: (0x2C13B74111C1F7E9, 33013481),
        2: (0xB5A76027F5D6AEB0, 16114352),
        4799: (0x53250DB90DEF10A4, 32194980),
        23999999: (0xB1CD0960FFA82E06, 5651462),
    }
    for lid, (h, slot) in vec.items():
        got = orc.fasthash64(struct.pack("<I", lid))
        assert got == h
        assert got % 36000000 == slot


def test_kat1_fasthash64_u64():
    vec = {0: (0x16C38EE185750EBC, 5), 1: (0xD7C65C9D6F0F512E, 53), 0x100000001: (0x3D0A67057705EBBC, 15)}
    for key, (h, bloom) in vec.items():
        got = orc.fasthash64(struct.pack("<Q", key))
        assert got == h and got >> 58 == bloom


def test_kat1_fastrand():
    import ctypes as C

    seed = C.c_uint64(0xDEADBEEF)
    L = orc.lib()
    L.orc_fastrand.restype = C.c_uint32
    L.orc_fastrand.argtypes = [C.POINTER(C.c_uint64)]
    assert [L.orc_fastrand(C.byref(seed)) for _ in range(3)] == [959880212, 3531117287, 3366701480]


def test_kat1_sizes():
    assert wire.FASST_MSG.itemsize == 9 and wire.TPL_MSG.itemsize == 6
    assert wire.STORE_MSG.itemsize == 53 and wire.TATP_MSG.itemsize == 55 and wire.SB_MSG.itemsize == 23


def test_kat2_lock_fasst():
    # KAT-2: lid=7: ACQ,ACQ,READ,COMMIT,READ,ACQ,ABORT,ACQ -> 5,6,4,8,4,5,7,5; READ vers 0 then 1
    F = wire.Fasst
    ops = [F.ACQUIRE_LOCK, F.ACQUIRE_LOCK, F.READ, F.COMMIT, F.READ, F.ACQUIRE_LOCK, F.ABORT, F.ACQUIRE_LOCK]
    m = np.zeros(len(ops), wire.FASST_MSG)
    m["type"] = ops
    m["lid"] = 7
    o = orc.FasstOracle(36_000_000)
    r = o.replay(m)
    assert r["type"].tolist() == [5, 6, 4, 8, 4, 5, 7, 5]
    assert r["ver"][2] == 0 and r["ver"][4] == 1
    assert (r["lid"] == 7).all()
    assert o.errors == 0


def test_kat3_store():
    # KAT-3: key 0x0000000100000000 (s_id 0, sf_type 1, start_time 0); val filled 0xEE, ver 0x11223344
    S = wire.Store
    o = orc.StoreOracle(2_000_000 * 18 // 4, 1000)  # only the first rows matter
    key = 0x0000000100000000

    def req(t, k, v0=None):
        m = np.zeros(1, wire.STORE_MSG)
        m["type"], m["key"], m["ver"] = t, k, 0x11223344
        m["val"][:] = 0xEE
        if v0 is not None:
            m["val"][0, 0] = v0
        return m

    r = o.replay(req(S.READ, key))
    assert r["type"][0] == 3 and r["val"][0, 0] == 21 and r["val"][0, 1] == 0x5A and r["ver"][0] == 0
    r = o.replay(req(S.SET, key, 7))
    assert r["type"][0] == 5 and r["ver"][0] == 0x11223344 and r["val"][0, 0] == 7 and r["val"][0, 1] == 0xEE
    r = o.replay(req(S.READ, key))
    assert r["val"][0, 0] == 7 and r["val"][0, 2] == 0xEE and r["ver"][0] == 1
    r = o.replay(req(S.SET, key, 9))
    assert r["type"][0] == 5
    r = o.replay(req(S.READ, key))
    assert r["val"][0, 0] == 9 and r["ver"][0] == 2
    q = req(S.READ, 0xDEAD000000000000)
    r = o.replay(q)
    assert r["type"][0] == 7
    q["type"] = 7
    assert r.tobytes() == q.tobytes()  # all other bytes echoed
    r = o.replay(req(S.SET, 0xDEAD000000000000))
    assert r["type"][0] == 7


def test_tatp_populate_magics():
    # in-band fixtures: tatp/udp/tatp.h:67-72 magic bytes at fixed value offsets
    o = orc.TatpOracle(2000)
    k, v, vals = o.dump(0)
    assert len(k) == 2000 and (vals[:, 32] == 97).all() and (v == 0).all()
    k, v, vals = o.dump(1)
    assert len(k) == 2000 and (vals[:, 4] == 98).all()
    k, v, vals = o.dump(2)
    assert 2000 <= len(k) <= 8000 and (vals[:, 0] == 99).all()
    k, v, vals = o.dump(3)
    assert 2000 <= len(k) <= 8000 and (vals[:, 3] == 100).all()
    assert 0.80 < vals[:, 0].mean() < 0.90  # is_active 85 %
    n_sf = len(k)
    k, v, vals = o.dump(4)
    assert (vals[:, 1] == 101).all() and 0.4 < len(k) / (3 * n_sf) < 0.6
    assert ((vals[:, 0] >= 1) & (vals[:, 0] <= 24)).all()


def test_smallbank_populate():
    o = orc.SmallbankOracle(1000)
    for t, magic in ((0, 97), (1, 98)):
        k, v, vals = o.dump(t)
        assert sorted(k.tolist()) == list(range(1000))
        assert (vals[:, 0] == magic).all()
        assert (vals[:, 4:8].view("<f4") == 1e9).all()
