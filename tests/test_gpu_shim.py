"""The UDP host shim (dint_amd/csrc/udp_shim.c) end to end on the GPU box: real datagrams in the
reference's wire format over loopback, replies compared with the CPU oracle; plus the port+1
CPU-usage echo the reference clients query at the end of a run."""
import os
import signal
import socket
import struct
import subprocess
import time

import numpy as np
import pytest

import tracegen
from dint_amd import wire
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERVER = os.path.join(ROOT, "dint_amd", "dint_udp_server")


def _free_udp_port():
    while True:
        s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        if p % 2 == 0 and p < 65000:
            return p


class Server:
    def __init__(self, *args):
        self.port = _free_udp_port()
        self.p = subprocess.Popen([SERVER, "--bind", "127.0.0.1", "--port", str(self.port), *args],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = self.p.stdout.readline()
        assert "ready" in line, (line, self.p.stderr.read() if self.p.poll() is not None else "")

    def stop(self):
        self.p.send_signal(signal.SIGTERM)
        out, _ = self.p.communicate(timeout=20)
        return out


def _exchange(port, req, window):
    """Closed loop over one socket: `window` datagrams out, `window` replies back, in order."""
    c = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    c.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 8 << 20)
    c.settimeout(10)
    size = req.dtype.itemsize
    raw = req.tobytes()
    out = bytearray()
    for lo in range(0, len(req), window):
        hi = min(len(req), lo + window)
        for i in range(lo, hi):
            c.sendto(raw[i * size:(i + 1) * size], ("127.0.0.1", port))
        for _ in range(lo, hi):
            d, _ = c.recvfrom(256)
            assert len(d) == size
            out += d
    c.close()
    return np.frombuffer(bytes(out), req.dtype)


def test_shim_lock_fasst_over_loopback():
    srv = Server("--workload", "fasst", "--slots", str(1 << 20), "--batch", "512", "--deadline-us", "200")
    try:
        req = tracegen.fasst_random(20_000, seed=41, n_hot=8, p_hot=0.8)
        got = _exchange(srv.port, req, 300)
        assert got.tobytes() == orc.FasstOracle(1 << 20).replay(req).tobytes()
        # a datagram of the wrong size is dropped, the server keeps serving
        c = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        c.sendto(b"\x00" * 5, ("127.0.0.1", srv.port))
        c.close()
        more = tracegen.fasst_random(100, seed=42)
        o = orc.FasstOracle(1 << 20)
        o.replay(req)
        assert _exchange(srv.port, more, 50).tobytes() == o.replay(more).tobytes()
    finally:
        out = srv.stop()
    assert "requests=20100" in out and "dropped=1" in out


def test_shim_tatp_over_loopback_and_cpu_monitor():
    n_sub, touch = 2000, 40
    srv = Server("--workload", "tatp", "--rows", str(n_sub), "--populate", str(touch), "--batch", "1024")
    try:
        o = orc.TatpOracle(n_sub, populate_n=touch)
        req = tracegen.tatp_random(15_000, [o.dump(t)[0] for t in range(5)], seed=43, n_sub_touch=touch)
        got = _exchange(srv.port, req, 500)
        assert got.tobytes() == o.replay(req).tobytes()
        # port + 1: 16 bytes in, {double ucores, double kcores} out (tatp/caladan/client_udp_shard.cc:75-92)
        c = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        c.settimeout(5)
        c.sendto(b"\x00" * 16, ("127.0.0.1", srv.port + 1))
        d, _ = c.recvfrom(64)
        c.close()
        u, k = struct.unpack("<dd", d)
        assert len(d) == 16 and 0 <= u < 256 and 0 <= k < 256
    finally:
        srv.stop()


def test_shim_sheds_load_with_the_ebpf_refusal_codes():
    """--shed: a batch that closes while the thread's other batch is still on the GPU is answered REJECT_READ at once
    (store/ebpf/store_kern.c:57-62); the client sends those requests again, as the reference's eBPF clients do."""
    S = wire.Store
    n_sub = 2000
    srv = Server("--workload", "store", "--rows", str(n_sub), "--populate", "40", "--batch", "256", "--deadline-us", "0",
                 "--threads", "1", "--shed")
    try:
        o = orc.StoreOracle(n_sub * 18 // 4, 40)
        req = tracegen.store_random(30_000, seed=3, n_sub_touch=40, p_set=0.0, p_missing=0.2)
        assert (req["type"] == S.READ).all()
        want = o.replay(req)
        c = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        c.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 32 << 20)
        c.settimeout(10)
        size = req.dtype.itemsize
        raw = req.tobytes()
        # `ver` is echoed on NOT_EXIST and overwritten on GRANT_READ: carry the request index in val[8:12] instead
        idx = np.arange(len(req), dtype="<u4")
        tagged = req.copy()
        tagged["val"][:, 8:12] = idx.view(np.uint8).reshape(-1, 4)
        raw = tagged.tobytes()
        got = {}
        refused = 0
        pending = list(range(len(req)))
        while pending:
            burst, pending = pending[:2000], pending[2000:]
            for i in burst:
                c.sendto(raw[i * size:(i + 1) * size], ("127.0.0.1", srv.port))
            for _ in burst:
                d, _ = c.recvfrom(256)
                r = np.frombuffer(d, wire.STORE_MSG)[0]
                if r["type"] == S.REJECT_READ:
                    refused += 1
                    i = int(np.frombuffer(r["val"][8:12].tobytes(), "<u4")[0])
                    assert d == raw[i * size:i * size + 1].replace(bytes([S.READ]), bytes([S.REJECT_READ])) + raw[i * size + 1:(i + 1) * size]
                    pending.append(i)  # send it again
                else:
                    # a served READ: GRANT_READ overwrites val (the tag is gone) -> match by key and position in `want`
                    got.setdefault(int(r["key"]), []).append(bytes(d))
        c.close()
        # every request was eventually served, with the oracle's answer (reads are idempotent; the table never changes)
        served = sum(len(v) for v in got.values())
        assert served == len(req)
        for i in range(0, len(req), 97):
            k = int(req["key"][i])
            w = want[i]
            if w["type"] == S.GRANT_READ:
                assert any(np.frombuffer(x, wire.STORE_MSG)[0]["val"].tobytes() == w["val"].tobytes() for x in got[k])
            else:
                assert any(np.frombuffer(x, wire.STORE_MSG)[0]["type"] == S.NOT_EXIST for x in got[k])
    finally:
        out = srv.stop()
    assert f"refused={refused}" in out


def test_shim_caladan_port_handshake():
    """--caladan: the control port answers net_req {int nports} with net_resp {int nports; u16 ports[]} and serves
    every data port (lock_fasst/caladan/server.cc:93-132, proto.h:38-45; the client side is
    lock_fasst/caladan/client_caladan.cc:285-310: one ClientLoop per returned port).  Replies leave from the port their
    request was sent to, and the three ports see one serial lock table."""
    srv = Server("--workload", "fasst", "--slots", str(1 << 20), "--batch", "256", "--deadline-us", "200", "--threads", "2",
                 "--caladan")
    try:
        ctl = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        ctl.settimeout(5)
        ctl.sendto(struct.pack("<i", 3), ("127.0.0.1", srv.port))
        d, frm = ctl.recvfrom(2048)
        assert frm[1] == srv.port and len(d) == 4 + 2 * 3
        nports, *ports = struct.unpack("<i3H", d)
        assert nports == 3 and len(set(ports)) == 3 and all(p not in (0, srv.port) for p in ports)
        ctl.sendto(struct.pack("<i", 2), ("127.0.0.1", srv.port))  # a second client machine gets ports of its own
        d2, _ = ctl.recvfrom(2048)
        more = struct.unpack("<i2H", d2)[1:]
        assert not set(more) & set(ports)
        ctl.close()
        # one closed-loop client per data port, as the reference's client threads: each sees its own replies, from its port
        req = tracegen.fasst_random(6_000, seed=44, n_hot=8, p_hot=0.8)
        o = orc.FasstOracle(1 << 20)
        socks = []
        for p in ports:
            c = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
            c.settimeout(10)
            socks.append((c, p))
        size = req.dtype.itemsize
        raw = req.tobytes()
        for i in range(len(req)):  # one outstanding request in all: the serial order is the send order
            c, p = socks[i % 3]
            c.sendto(raw[i * size:(i + 1) * size], ("127.0.0.1", p))
            d, frm = c.recvfrom(64)
            assert frm[1] == p
            assert d == o.replay(req[i:i + 1]).tobytes(), i
        # and a burst: 200 in flight per port
        burst = tracegen.fasst_random(600, seed=45, n_hot=4, p_hot=0.0)  # distinct cold lids: order between ports is irrelevant
        braw = burst.tobytes()
        for i in range(len(burst)):
            socks[i % 3][0].sendto(braw[i * size:(i + 1) * size], ("127.0.0.1", socks[i % 3][1]))
        got = sum(1 for k in range(len(burst)) if len(socks[k % 3][0].recvfrom(64)[0]) == size)
        assert got == len(burst)
        for c, _ in socks:
            c.close()
    finally:
        out = srv.stop()
    assert "requests=6600" in out


def _n_sockets(pid):
    n = 0
    for f in os.listdir(f"/proc/{pid}/fd"):
        try:
            n += os.readlink(f"/proc/{pid}/fd/{f}").startswith("socket:")
        except OSError:
            pass
    return n


def test_shim_caladan_data_sockets_are_bounded_and_reaped():
    """every handshake opens up to 730 data sockets.  The shim must (a) refuse -- not half serve -- a
    handshake that does not fit under the process's file limit, and keep serving smaller ones; (b) close data sockets
    nobody has sent to for --idle-s seconds, so that clients coming and going do not exhaust the limit."""
    import resource

    def limit():
        resource.setrlimit(resource.RLIMIT_NOFILE, (400, 400))

    port = _free_udp_port()
    p = subprocess.Popen([SERVER, "--bind", "127.0.0.1", "--port", str(port), "--workload", "fasst", "--slots", "65536",
                          "--batch", "64", "--threads", "2", "--caladan", "--idle-s", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, preexec_fn=limit)
    try:
        assert "ready" in p.stdout.readline()
        base = _n_sockets(p.pid)
        ctl = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        ctl.settimeout(1.0)
        ctl.sendto(struct.pack("<i", 500), ("127.0.0.1", port))  # more than the limit leaves room for
        with pytest.raises(socket.timeout):
            ctl.recvfrom(4096)
        assert _n_sockets(p.pid) == base  # nothing half-opened stays behind
        ctl.settimeout(5)
        ctl.sendto(struct.pack("<i", 40), ("127.0.0.1", port))
        d, _ = ctl.recvfrom(4096)
        ports = struct.unpack("<i40H", d)[1:]
        assert len(set(ports)) == 40 and _n_sockets(p.pid) == base + 40
        # keep ONE of them busy; the 39 others are idle and must be gone within a few seconds
        c = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        c.settimeout(5)
        req = tracegen.fasst_random(1, seed=3, n_hot=1, p_hot=0.0).tobytes()
        t_end = time.time() + 4.0
        while time.time() < t_end:
            c.sendto(req, ("127.0.0.1", ports[7]))
            assert len(c.recvfrom(64)[0]) == len(req)
            time.sleep(0.2)
        assert _n_sockets(p.pid) == base + 1
        # and the room they gave back can be handed out again, over and over
        for _ in range(3):
            ctl.sendto(struct.pack("<i", 300), ("127.0.0.1", port))
            d, _ = ctl.recvfrom(4096)
            assert struct.unpack("<i", d[:4])[0] == 300
            time.sleep(2.5)
    finally:
        p.send_signal(signal.SIGTERM)
        p.communicate(timeout=20)
