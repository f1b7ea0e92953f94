"""Rebuild a replica from a drained log.

The reference replicates every committed write three ways -- COMMIT_LOG / DELETE_LOG to all three servers, then the
backups, then the primary (tatp/caladan/client_udp_shard.cc:486-570) -- but never reads a log back: there is no
recovery path.  With the log in HBM and `dint_log_drain` streaming it out, recovery is a replay: every
log record becomes the backup operation the client sent right after it (COMMIT_BCK for a row that exists, INSERT_BCK for
one that does not yet, DELETE_BCK for a DELETE_LOG record), in log order, through the ordinary hot path of the
replica being rebuilt.  Because COMMIT_BCK bumps the version and INSERT_BCK starts at 0, a replica that starts from
the same image as the logging server (e.g. the initial population) ends with identical rows AND versions.
"""
from __future__ import annotations

import numpy as np

from .wire import SB_MSG, TATP_MSG, Sb, Tatp, Workload


def apply_log(engine, records: np.ndarray) -> dict:
    """records: LOG_REC array as returned by Engine.log_drain / read_log, oldest first."""
    n = len(records)
    if n == 0:
        return {"applied": 0}
    if engine.workload == Workload.SMALLBANK:  # no inserts / deletes: every record is a COMMIT_BCK
        m = np.zeros(n, SB_MSG)
        m["type"], m["table"], m["key"], m["ver"] = Sb.COMMIT_BCK, records["table"], records["key"], records["ver"]
        m["val"] = records["val"][:, :8]
        rep = engine.submit(m)
        return {"applied": n, "acks": int((rep["type"] == Sb.COMMIT_BCK_ACK).sum())}
    assert engine.workload == Workload.TATP
    # which rows exist before the replay: one READ per distinct (table, key)
    tk = (records["table"].astype(np.uint64) << np.uint64(60)) ^ records["key"]  # keys use < 48 bits
    uniq, first, inv = np.unique(tk, return_index=True, return_inverse=True)
    rd = np.zeros(len(uniq), TATP_MSG)
    rd["type"], rd["table"], rd["key"] = Tatp.READ, records["table"][first], records["key"][first]
    exists0 = engine.submit(rd)["type"] == Tatp.GRANT_READ
    # existence before record i = what the previous record on the same row left, else the initial state
    order = np.argsort(inv, kind="stable")
    same_as_prev = np.zeros(n, bool)
    same_as_prev[order[1:]] = inv[order[1:]] == inv[order[:-1]]
    prev = np.empty(n, np.int64)
    prev[order[1:]] = order[:-1]
    prev[order[0]] = order[0]
    is_del = records["is_del"] != 0
    exists = np.where(same_as_prev, ~is_del[prev], exists0[inv])
    m = np.zeros(n, TATP_MSG)
    m["table"], m["key"], m["val"], m["ver"] = records["table"], records["key"], records["val"], records["ver"]
    m["type"] = np.where(is_del, Tatp.DELETE_BCK, np.where(exists, Tatp.COMMIT_BCK, Tatp.INSERT_BCK))
    rep = engine.submit(m)
    return {"applied": n, "commits": int((rep["type"] == Tatp.COMMIT_BCK_ACK).sum()),
            "inserts": int((rep["type"] == Tatp.INSERT_BCK_ACK).sum()), "deletes": int((rep["type"] == Tatp.DELETE_BCK_ACK).sum())}
