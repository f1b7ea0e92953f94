"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and
exports every symbol include/dint_abi.h declares; the product path has no CPU fallback."""
import ctypes as C
import os
import re

import pytest

from dint_amd import _lib, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "dint_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dint_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in dint_abi.h but not exported by libdint.so"
    assert sorted(_lib.SYMBOLS) == syms


def test_msg_sizes_match_wire_structs():
    L = _lib.load()
    for wl, dt in wire.MSG_DTYPE.items():
        assert L.dint_msg_size(int(wl)) == dt.itemsize
    assert L.dint_msg_size(99) < 0


def test_no_cpu_fallback():
    """Without a GPU the engine must refuse to come up (it never computes on the CPU)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dint_amd.engine import Engine

    with pytest.raises(_lib.DintError):
        Engine(wire.Workload.FASST, n_slots=1024)


def test_config_struct_layout():
    assert C.sizeof(_lib.Config) == 4 * 4 + 8 * 2 + 4 * 3 + 4 * 5
    assert C.sizeof(_lib.Stats) == 8 * 12  # ABI v4: + late_requests, reserved[3]


def test_flag_constants_match_the_header():
    """the Python mirror of dint_config.flags (dint_amd._lib.FLAG_*) against include/dint_abi.h"""
    src = open(os.path.join(ROOT, "include", "dint_abi.h")).read()
    hdr = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define DINT_FLAG_([A-Z_]+) (\d+)u", src)}
    assert hdr == {"KV_ROUNDS": _lib.FLAG_KV_ROUNDS, "COPY_STREAMS": _lib.FLAG_COPY_STREAMS, "LOCK_SAME_KEY": _lib.FLAG_LOCK_SAME_KEY,
                   "KV_NO_HOT": _lib.FLAG_KV_NO_HOT, "INPUTS_READY": _lib.FLAG_INPUTS_READY}
    assert len(set(hdr.values())) == len(hdr) and all(v & (v - 1) == 0 for v in hdr.values())  # distinct single bits


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "dint_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".c")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_segments_multi_rejects_bad_arguments_without_a_device():
    """dint_submit_segments_multi checks its items before it touches a device: no items, a null engine"""
    L = _lib.load()
    assert L.dint_submit_segments_multi(None, 0, None) < 0
    items = (_lib.SegmentsItem * 2)()
    assert L.dint_submit_segments_multi(items, 2, None) < 0  # engine == NULL
    assert b"null" in L.dint_last_error()
    assert C.sizeof(_lib.SegmentsItem) == 48  # {engine, d_base, n_seg, seg_cap, seg_stride, d_cnt, cnt_stride}
