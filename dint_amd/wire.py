"""Wire formats of the DINT server path: the packed request/reply structs and opcodes.

These are the reference's ``#pragma pack(1)`` structs, byte for byte, as numpy
structured dtypes (``align=False``), so a batch is a contiguous ``N x sizeof``
byte array exactly as N UDP payloads laid end to end.

Reference definitions (relative to the reference tree):
  lock_fasst/udp/net.h:11-29   message{type, lid, ver}                 9 B
  lock_2pl/udp/net.h:11-31     message{action, lid, type}              6 B
  log_server/udp/net.h:15-30   message{type, key, val[40], ver}       53 B
  store/udp/net.h:15-41        message{type, key, val[40], ver}       53 B
  tatp/udp/net.h:15-66         message{ord, type, table, key, val[40], ver}  55 B
  smallbank/udp/net.h:15-50    message{ord, type, table, key, val[8], ver}   23 B
"""
from __future__ import annotations

import enum

import numpy as np

FASST_MSG = np.dtype([("type", "u1"), ("lid", "<u4"), ("ver", "<u4")], align=False)
TPL_MSG = np.dtype([("action", "u1"), ("lid", "<u4"), ("type", "u1")], align=False)
LOG_MSG = np.dtype([("type", "u1"), ("key", "<u8"), ("val", "u1", (40,)), ("ver", "<u4")], align=False)
STORE_MSG = LOG_MSG
TATP_MSG = np.dtype(
    [("ord", "u1"), ("type", "u1"), ("table", "u1"), ("key", "<u8"), ("val", "u1", (40,)), ("ver", "<u4")],
    align=False,
)
SB_MSG = np.dtype(
    [("ord", "u1"), ("type", "u1"), ("table", "u1"), ("key", "<u8"), ("val", "u1", (8,)), ("ver", "<u4")],
    align=False,
)

assert FASST_MSG.itemsize == 9 and TPL_MSG.itemsize == 6 and LOG_MSG.itemsize == 53
assert TATP_MSG.itemsize == 55 and SB_MSG.itemsize == 23

#: canonical 64-byte log record every engine log ring uses
LOG_REC = np.dtype(
    [("key", "<u8"), ("val", "u1", (40,)), ("ver", "<u4"), ("is_del", "u1"), ("table", "u1"), ("pad", "u1", (10,))],
    align=False,
)
assert LOG_REC.itemsize == 64


class Workload(enum.IntEnum):
    """Engine flavours; values are the ``workload`` field of ``dint_config``."""

    FASST = 0
    TPL = 1  # lock_2pl
    LOG = 2
    STORE = 3
    TATP = 4
    SMALLBANK = 5


MSG_DTYPE = {
    Workload.FASST: FASST_MSG,
    Workload.TPL: TPL_MSG,
    Workload.LOG: LOG_MSG,
    Workload.STORE: STORE_MSG,
    Workload.TATP: TATP_MSG,
    Workload.SMALLBANK: SB_MSG,
}


class Fasst(enum.IntEnum):  # lock_fasst/udp/net.h:11-21
    READ = 0
    ACQUIRE_LOCK = 1
    ABORT = 2
    COMMIT = 3
    GRANT_READ = 4
    GRANT_LOCK = 5
    REJECT_LOCK = 6
    ABORT_ACK = 7
    COMMIT_ACK = 8


class Tpl(enum.IntEnum):  # lock_2pl/udp/net.h:11-23
    ACQUIRE_LOCK = 0
    RELEASE_LOCK = 1
    GRANT_LOCK = 2
    REJECT_LOCK = 3
    RETRY = 4
    RELEASE_ACK = 5
    SHARED = 0
    EXCLUSIVE = 1


class Log(enum.IntEnum):  # log_server/udp/net.h:15-18
    COMMIT = 0
    ACK = 1


class Store(enum.IntEnum):  # store/udp/net.h:15-29
    READ = 0
    SET = 1
    INSERT = 2
    GRANT_READ = 3
    REJECT_READ = 4
    SET_ACK = 5
    REJECT_SET = 6
    NOT_EXIST = 7
    INSERT_ACK = 8
    REJECT_INSERT = 9


class Tatp(enum.IntEnum):  # tatp/udp/net.h:15-52
    READ = 0
    ACQUIRE_LOCK = 1
    ABORT = 2
    COMMIT = 3
    GRANT_READ = 4
    REJECT_READ = 5
    NOT_EXIST = 6
    GRANT_LOCK = 7
    REJECT_LOCK = 8
    ABORT_ACK = 9
    COMMIT_ACK = 10
    REJECT_COMMIT = 11
    COMMIT_PRIM = 12
    COMMIT_BCK = 13
    COMMIT_LOG = 14
    COMMIT_PRIM_ACK = 15
    COMMIT_BCK_ACK = 16
    COMMIT_LOG_ACK = 17
    INSERT_PRIM = 18
    INSERT_BCK = 19
    INSERT_PRIM_ACK = 20
    INSERT_BCK_ACK = 21
    DELETE_PRIM = 22
    DELETE_BCK = 23
    DELETE_LOG = 24
    DELETE_PRIM_ACK = 25
    DELETE_BCK_ACK = 26
    DELETE_LOG_ACK = 27


class TatpTable(enum.IntEnum):  # tatp/udp/kvs.h:10-17
    SUBSCRIBER = 0
    SECOND_SUBSCRIBER = 1
    ACCESS_INFO = 2
    SPECIAL_FACILITY = 3
    CALL_FORWARDING = 4


class Sb(enum.IntEnum):  # smallbank/udp/net.h:15-38
    ACQUIRE_SHARED = 0
    ACQUIRE_EXCLUSIVE = 1
    RELEASE_SHARED = 2
    RELEASE_EXCLUSIVE = 3
    COMMIT_PRIM = 4
    COMMIT_BCK = 5
    COMMIT_LOG = 6
    GRANT_SHARED = 7
    REJECT_SHARED = 8
    GRANT_EXCLUSIVE = 9
    REJECT_EXCLUSIVE = 10
    RELEASE_SHARED_ACK = 11
    RELEASE_EXCLUSIVE_ACK = 12
    COMMIT_PRIM_ACK = 13
    COMMIT_BCK_ACK = 14
    COMMIT_LOG_ACK = 15
    RETRY = 16
    WARMUP_READ = 17
    WARMUP_READ_ACK = 18


class SbTable(enum.IntEnum):  # smallbank/udp/kvs.h:10-14
    SAVING = 0
    CHECKING = 1
