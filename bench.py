#!/usr/bin/env python3
"""bench.py -- the DINT server hot path on MI355X, one step = one batch of synthetic wire messages
already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tatp|smallbank|store|fasst] ...

Default workload (BASELINE.json metric "Mtxn/s ... TATP Zipf-0.8", configs[3]): the TATP transaction
mix against 1M subscribers.  W virtual clients run the reference client's seven transactions in lock
step (dint_amd/csrc/txn_driver.cc); one step = one EPOCH = every client's current phase, i.e. three
request batches (one per replicated shard server, as in the reference's 3-server deployment), which the
three engines of the GPU process.  The closed loop is first run once and recorded; the timed region
replays the recorded batches from HBM (no host work, no PCIe) and the replies are checked byte for byte
against the recorded run -- and, in the cpu_baseline leg, against the CPU oracle on the same stream.
Other workloads: `smallbank` (BASELINE configs[4], same machinery), `store` (configs[2]: 16.8M keys, 95/5
read/write), `fasst` (configs[1]: lock_fasst, 64k batches; `--slots 36000000` = the reference's table).

`--gpus N` with N > 1 and no torchrun environment re-launches this file under torch.distributed.run, one rank
per GPU (RCCL); if fewer than N GPUs are visible the ranks share them and the exchange is staged through the
host over gloo -- a functional run, labelled as such.  Rank 0 prints ONE JSON line; DESIGN.md "Measurement"
defines every field.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (GPU_MAX_HW_QUEUES is left at HIP's default of 4: raising it to 8 / 16 so that the exchange stream and RCCL's stream get
# queues of their own HALVED the pipelined-exchange rate on MI355X -- r02, NOTEBOOK.md: 0.49 -> 1.04 ms per epoch.)

BATCH = 65536
KV_PASS = 1 << 20  # requests per kernel pass of the store / tatp / smallbank engines (dint_config.max_pass = 0)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("DINT_BENCH_WORKLOAD", "tatp"),
                    choices=["tatp", "smallbank", "store", "fasst", "2pl", "log"])
    ap.add_argument("--per-step", type=int, default=16,
                    help="batches (tatp / smallbank: epochs) per step: a step is a fixed bundle, so that K steps are "
                         "tens of milliseconds of GPU work, not a few launches")
    ap.add_argument("--slots", type=int, default=1 << 20, help="lock_fasst table slots (BASELINE configs[1]: 1M; reference 36000000)")
    ap.add_argument("--theta", type=float, default=None,
                    help="Zipf skew of the key stream (default 0.8; smallbank 0.99); 0 = the reference's own distribution")
    ap.add_argument("--subscribers", type=int, default=1_000_000, help="tatp subscribers PER GPU (BASELINE configs[3]: 1M on one GPU)")
    ap.add_argument("--accounts", type=int, default=0,
                    help="smallbank accounts in total over all GPUs (default 10M per GPU: BASELINE configs[4] = 80M on 8)")
    ap.add_argument("--keys", type=int, default=16_777_216, help="store keys (BASELINE configs[2]: 16M)")
    ap.add_argument("--clients", type=int, default=524288,
                    help="closed-loop clients per GPU (one outstanding phase each: a tatp shard server sees ~clients * 0.46 "
                         "requests per epoch, all resolved in one kernel pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-reference", action="store_true",
                    help="tatp: skip the unmodified reference server leg (7M subscribers on both sides; its populate takes minutes)")
    ap.add_argument("--ref-timeout", type=float, default=300.0,
                    help="seconds to wait for the reference server's populate once the GPU legs are done")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the GPU-resident closed-loop leg")
    ap.add_argument("--no-rand64", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive dint_submit_async measurement")
    ap.add_argument("--force-exchange", action="store_true", help="N = 1: still run the multi-GPU exchange (self all-to-all)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="tatp: skip the compact legs of the other BASELINE configs (lock_fasst, lock_2pl, log, store, smallbank)")
    ap.add_argument("--no-shim", action="store_true", help="skip the UDP shim loopback leg")
    ap.add_argument("--no-mixes", action="store_true", help="store: skip the 100/0 and 50/50 legs (profile runs)")
    ap.add_argument("--no-exchange-leg", action="store_true", help="skip the compact --force-exchange leg of the default line")
    ap.add_argument("--no-as-shipped", action="store_true", help="skip the as-shipped tatp/udp deployment leg (three reference "
                    "server processes, ~27 GB of host memory)")
    ap.add_argument("--legs", default="all", choices=["all", "gpu", "headline"],
                    help="one switch over the --no-* flags: all = every leg (the driver's run); gpu = no CPU leg (cpu baseline, "
                         "reference, as-shipped servers, UDP shim); headline = the timed region, its kernel times and roofline "
                         "only (what tools/profile_bench.py runs under rocprofv3)")
    ap.add_argument("--replay", default="inplace", choices=["inplace", "copy"],
                    help="tatp / smallbank / store replay: inplace = every batch answered in place, as the reference answers a "
                         "datagram (the receive buffers are refilled from pristine copies outside the timed region: the NIC's DMA); "
                         "copy = separate reply buffers (r01-r05: the partition kernel copies request -> reply)")
    ap.add_argument("--no-ahead", action="store_true",
                    help="tatp / store replay: plain dint_submit_device calls (r05), no dint_submit_device_ahead")
    ap.add_argument("--sweep-clients", action="store_true",
                    help="tatp / smallbank: abort rate and Mtxn/s at 4096 / 32768 / 131072 / 524288 closed-loop clients")
    args = ap.parse_args()
    if args.legs in ("gpu", "headline"):
        args.no_cpu_baseline = args.no_cpu_reference = args.no_shim = args.no_as_shipped = True
    args.no_pass_1m = args.no_inputs_ready = False
    if args.legs == "headline":
        # (lock tables: the DINT_FLAG_INPUTS_READY leg is a side leg like the others; a profile is of the timed region's kernels)
        args.no_inputs_ready = True
        args.no_closed_loop = args.no_rand64 = args.no_host_path = args.no_other_workloads = args.no_mixes = args.no_exchange_leg = True
        args.no_pass_1m = True  # (lock tables: the 2^20-request passes would fall into the profile's "last N dispatches")
    return args


# --------------------------------------------------------------------------------------------- launch / dist
def respawn_under_torchrun(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start N ranks of this same file."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def init_dist(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    transport = "self"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank % ndev)
        if ndev >= world:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            transport = "nccl"
        else:  # several ranks per GPU: RCCL refuses that; functional run over gloo, exchange staged through the host
            dist.init_process_group("gloo")
            transport = "host"
    else:
        torch.cuda.set_device(0)
    return world, rank, torch.cuda.current_device(), transport


def barrier(world):
    import torch
    import torch.distributed as dist

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def _reduce(x, world, op, transport):
    import torch
    import torch.distributed as dist

    if world > 1:
        t = torch.tensor([float(x)], device="cuda" if transport == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=op)
        x = float(t.item())
    return x


def max_over_ranks(dt, world, transport="nccl"):
    import torch.distributed as dist

    return _reduce(dt, world, dist.ReduceOp.MAX, transport)


def sum_over_ranks(x, world, transport="nccl"):
    import torch.distributed as dist

    return _reduce(x, world, dist.ReduceOp.SUM, transport)


# --------------------------------------------------------------------------------------------- helpers
def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "logical_cores": os.cpu_count()}


KERNEL_SOURCES = {  # the files the device code of a workload's pass is compiled from (dint_amd/csrc/)
    "fasst": ("k_locks.hip", "dint_bins.h", "dint_device.h", "dint_kernels.h"),
    "2pl": ("k_locks.hip", "dint_bins.h", "dint_device.h", "dint_kernels.h"),
    "log": ("k_log.hip", "dint_device.h", "dint_kernels.h"),
    None: ("k_kv_dev.h", "k_kv.hip", "dint_bins.h", "dint_device.h", "dint_kernels.h", "dint_kv.h", "dint_kv_core.h"),
}


def kernel_source_hash(workload=None):
    """sha256 over the sources of the workload's pass kernels: profiles/ counters are only quoted for the build they
    were taken from (host code, the routing kernels and the client kernels do not enter)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dint_amd", "csrc")
    for f in KERNEL_SOURCES.get(workload, KERNEL_SOURCES[None]):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profile_counters(workload, kernels):
    """HBM-side bytes per launch of `kernels` (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes) and their
    rocprofv3 begin->end duration, from profiles/traffic_<workload>.json -- written by tools/profile_bench.py from
    rocprofv3 runs of this same command.  Quoted only when the file was taken from the kernel sources this run
    uses; otherwise (None, None, reason).  Not measured in this run: the field name in the JSON line says so."""
    path = os.path.join(ROOT, "profiles", f"traffic_{workload}.json")
    try:
        p = json.load(open(path))
    except (OSError, ValueError):
        return None
    if p.get("kernel_source_hash") != kernel_source_hash(workload):
        return {"stale": True, "file": os.path.relpath(path, ROOT), "profiled_sources": p.get("kernel_source_hash")}
    tot_b, tot_us, seen = 0.0, 0.0, 0
    for k in kernels:
        e = p.get("kernels", {}).get(k)
        if e is None:
            continue
        seen += 1
        tot_b += e.get("fetch_bytes", 0.0) + e.get("write_bytes", 0.0)
        tot_us += e.get("avg_us", 0.0)
    if not seen:
        return None
    return {"traffic_bytes": int(tot_b), "kernel_avg_us": round(tot_us, 3), "file": os.path.relpath(path, ROOT),
            "command": p.get("command"), "launches": p.get("launches"), "commit": p.get("commit")}


L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
_RAND_CACHE = {}


def rand_roofline(extra, ops_per_s, dev, U, u_what, table_gb):
    """The north star's fraction of the random-64-B HBM roofline: ops/s x U / (random sector gathers/s), U = the 64-byte
    table sectors a request touches at random by the layout (stated per workload; request / reply streams and the pass
    scratch are not in it).  The denominator is measured here (csrc/k_bench.hip): one 8-byte load from a random
    64-byte-aligned sector -- a sector costs the same whatever part of it is used -- over a table of the engines'
    own footprint and over 8 GB (past ~4 GB the 4 x 16-byte form of r02 was TLB-bound: 21 G/s), 8 workgroups per CU;
    the best rate is the roofline."""
    from dint_amd.engine import bench_access

    try:
        den = {}
        for gb in sorted({max(1, min(int(round(table_gb)), 16)), 8}):
            key = (dev, gb)
            if key not in _RAND_CACHE:
                _RAND_CACHE[key] = {"gather8": bench_access(gb << 30, 1 << 28, "gather", 8, 8, dev)[0],
                                    "gather64": bench_access(gb << 30, 1 << 28, "gather", 64, 8, dev)[0],
                                    "rmw64": bench_access(gb << 30, 1 << 28, "rmw", 64, 8, dev)[0],
                                    "scatter8": bench_access(gb << 30, 1 << 28, "scatter", 8, 8, dev)[0]}
            den[f"{gb}GB"] = {k: round(v / 1e9, 2) for k, v in _RAND_CACHE[key].items()}
        best = max(max(v["gather8"], v["gather64"]) for v in den.values()) * 1e9
        extra["roofline_rand64"] = {"frac": round(ops_per_s * U / best, 4), "U": round(U, 3), "U_what": u_what,
                                    "ops_per_s": round(ops_per_s), "gathers_per_s": round(best),
                                    "G_per_s_by_table_size": den,
                                    "what": "ops/s x U / measured random 64-B sector gathers/s (best of the table sizes)"}
    except Exception as ex:  # measurement helper only
        extra["roofline_rand64_error"] = str(ex)


def pct(a, q):
    return round(float(np.percentile(a, q)), 2)


# ----------------------------------------------------------------------------- lock_fasst / lock_2pl / log
def cpu_baseline_micro(kind, sample: np.ndarray, size: int, want: bytes):
    """The CPU baseline on rank 0's host cores over a bounded sample of the same stream: the unmodified reference
    server when its replay binary is present and the table has the reference's size (kind "reference"), else the C
    restatement.  Its replies must equal the GPU's on the same requests (`want`)."""
    from oracle import oracle as orc

    ref_name = {"fasst": "lock_fasst", "2pl": "lock_2pl", "log": "log_server"}[kind]
    ref_size = 1_000_000 if kind == "log" else 36_000_000
    if size == ref_size and orc.ref_available(ref_name):
        rep, st = orc.ref_replay(ref_name, sample)
        out = {"value": st["ops_per_s"] / 1e6, "unit": "Mtxn/s", "cores": 1, "kind": "reference",
               "sample": f"{len(sample)} requests of the bench stream, unmodified {ref_name}/udp/server.cc, sockets interposed"}
    else:
        o = {"fasst": orc.FasstOracle, "2pl": orc.TplOracle, "log": orc.LogOracle}[kind](size)
        t = time.perf_counter()
        rep = o.replay(sample)
        dt = time.perf_counter() - t
        out = {"value": len(sample) / dt / 1e6, "unit": "Mtxn/s", "cores": 1, "kind": "port",
               "sample": f"{len(sample)} requests of the bench stream, oracle/dint_oracle.c ({size} {'ring entries' if kind == 'log' else 'slots'})"}
    out["value"] = round(out["value"], 4)
    out["host_cpu"] = host_cpu()
    out["oracle_parity"] = {"requests": len(sample), "ok": rep.tobytes() == want}
    return out


def cpu_as_shipped_tatp(shipped, trace, ops_per_txn, args):
    """BASELINE.md 3(2) / configs[3] "compare ... vs tatp/udp CPU server": the reference's as-shipped deployment -- three
    unmodified `tatp/udp/server_shard <id> 8` processes (exp/run_tatp.sh; their own main(), populate at 7M subscribers,
    thread pinning, kernel UDP sockets; bind() redirected 10.10.1.N -> 127.0.1.N) -- each driven closed-loop over loopback
    with ITS stream of the recorded 7M-subscriber epochs, all three at once.  The closed-loop client replays requests out of
    transaction context and wraps around, so the requests the udp server panics on out of context are left out (a COMMIT /
    DELETE of a CALL_FORWARDING row that is not there: tatp/udp/kvs.h:91,152; INSERTs, which would grow the table with every
    lap): READ, ACQUIRE_LOCK, ABORT, the log appends and the commits of the three static tables remain (~97 % of the stream)."""
    try:
        if not shipped.wait_populated(args.ref_timeout):
            return {"error": f"the as-shipped servers were not populated within {args.ref_timeout:.0f} s"}
        streams, kept, total = [], 0, 0
        for sv in range(3):
            m = np.concatenate([trace[e][0][sv] for e in range(len(trace))])
            safe = np.isin(m["type"], (0, 1, 2, 14, 24)) | (np.isin(m["type"], (12, 13)) & (m["table"] < 4))
            kept, total = kept + int(safe.sum()), total + len(m)
            streams.append(m[safe].copy())
        cores = os.cpu_count() or 8
        r = shipped.run(streams, client_threads=min(16, max(2, cores // 8)), window=32, warmup_s=1.0, measure_s=4.0)
    except Exception as ex:
        return {"error": f"{type(ex).__name__}: {ex}"}
    return {"value": round(r["ops_per_s"] / ops_per_txn / 1e6, 4), "unit": "Mtxn/s", "cores": 3 * r["server_threads"], "kind": "reference",
            "path": "as shipped: three unmodified tatp/udp/server_shard.cc processes (8 worker threads each) over loopback UDP",
            "ops_per_s": round(r["ops_per_s"]), "ops_per_txn": round(ops_per_txn, 3), "lost": r["lost"], "populate_s": r["populate_s"],
            "servers": r["servers"], "host_cpu": host_cpu(),
            "sample": f"{kept} of {total} requests of the three servers' streams of {len(trace)} closed-loop epochs at 7,000,000 "
                      f"subscribers, replayed closed-loop for 4 s (16 client threads x 32 outstanding per server)"}


def cpu_as_shipped_fasst(sample: np.ndarray):
    """the reference's as-shipped `lock_fasst/udp/server T` (unmodified; real UDP sockets, on
    127.0.0.1) driven closed-loop over loopback by dint_amd/csrc/udp_loop_client.c with the bench stream: T = 8
    (exp/run_lock_fasst.sh:48-49) and T = a quarter of the host's logical cores.  The table has the reference's
    compile-time size (36M slots) whatever --slots says: this leg prices the reference's per-packet path (syscalls +
    kernel UDP stack + the lock op), not the table."""
    from oracle import oracle as orc

    if not orc.loopback_available():
        return None
    cores = os.cpu_count() or 8
    runs = []
    for t in sorted({8, max(8, cores // 4)}):
        try:
            r = orc.ref_loopback_fasst(sample, server_threads=t, client_threads=min(2 * t, max(4, cores // 2)), window=32,
                                       warmup_s=1.0, measure_s=4.0)
        except Exception as ex:
            r = {"server_threads": t, "error": f"{type(ex).__name__}: {ex}"}
        runs.append(r)
    ok = [r for r in runs if "ops_per_s" in r]
    if not ok:
        return {"runs": runs}
    best = max(ok, key=lambda r: r["ops_per_s"])
    return {"value": round(best["ops_per_s"] / 1e6, 4), "unit": "Mtxn/s", "cores": best["server_threads"],
            "kind": "reference", "path": "as shipped: unmodified lock_fasst/udp/server.cc over loopback UDP",
            "host_cpu": host_cpu(), "runs": runs,
            "sample": f"{len(sample)} requests of the bench stream replayed closed-loop for 4 s per run "
                      f"(client threads x window 32 outstanding requests each)"}


def shim_loopback(sample: np.ndarray, dev: int, threads=(2, 8, 16)):
    """SURVEY.md 8 f1, measured: the UDP host shim (dint_amd/dint_udp_server: recvmmsg -> dint_submit_async -> sendmmsg,
    the GPU engine behind it) over loopback, driven by the same closed-loop client as `cpu_as_shipped` with the same
    lock_fasst stream, for 2 / 8 / 16 socket threads.  Kernel UDP stack on both sides of every datagram: this is what
    a socket-fed deployment of the engine delivers on this host, next to the reference's own server."""
    import socket as _s
    import subprocess as sp
    import tempfile

    here = os.path.join(ROOT, "dint_amd")
    server, client = os.path.join(here, "dint_udp_server"), os.path.join(here, "dint_udp_client")
    if not (os.access(server, os.X_OK) and os.access(client, os.X_OK)):
        return None
    cores = os.cpu_count() or 8
    runs = []
    with tempfile.TemporaryDirectory(prefix="dint_shim_") as td:
        tp = os.path.join(td, "requests.bin")
        np.ascontiguousarray(sample).tofile(tp)
        for t in threads:
            so = _s.socket(_s.AF_INET, _s.SOCK_DGRAM)
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1] & ~1
            so.close()
            srv = sp.Popen([server, "--workload", "fasst", "--bind", "127.0.0.1", "--port", str(port), "--threads", str(t),
                            "--batch", "4096", "--deadline-us", "50", "--device", str(dev)], stdout=sp.PIPE, stderr=sp.DEVNULL, text=True)
            try:
                line = srv.stdout.readline()
                if "ready" not in line:
                    raise RuntimeError(f"shim did not start: {line!r}")
                res = sp.run([client, tp, str(sample.dtype.itemsize), str(port), str(min(4 * t, max(4, cores // 2))), "64", "1.0", "3.0"],
                             capture_output=True, text=True, timeout=60)
                r = json.loads(res.stdout.strip().splitlines()[-1])
                r["server_threads"] = t
            except Exception as ex:
                r = {"server_threads": t, "error": f"{type(ex).__name__}: {ex}"}
            finally:
                srv.terminate()  # the exact process started above
                try:
                    srv.wait(timeout=20)
                except sp.TimeoutExpired:
                    srv.kill()
            runs.append(r)
    ok = [r for r in runs if "ops_per_s" in r]
    if not ok:
        return {"runs": runs}
    best = max(ok, key=lambda r: r["ops_per_s"])
    return {"value": round(best["ops_per_s"] / 1e6, 4), "unit": "M requests/s", "socket_threads": best["server_threads"],
            "runs": runs, "path": "dint_udp_server (recvmmsg / dint_submit_async / sendmmsg) over loopback UDP, lock_fasst, 36M slots",
            "what": "closed-loop loopback clients, 64 outstanding requests per client thread, 3 s per run"}


def bench_lock(args, world, rank, dev, transport, kind):
    """lock_fasst (BASELINE configs[1]) and lock_2pl: the reference's load generators restated -- lock_fasst/caladan/
    client.cc:183-280 (csrc/fasst_client.cc) and lock_2pl/caladan/client.cc:167-240 (dint_amd.driver.TplClient) -- 4096
    closed-loop workers over 24M lids; the trace is generated once through the engine itself (the clients need the
    replies) and cut into 64k-request batches, which the timed region replays from HBM.  One step = `--per-step`
    batches."""
    import torch

    from dint_amd import wire
    from dint_amd.engine import Engine
    from dint_amd.sharded import Router

    K, W, B = args.steps, args.warmup, args.per_step
    theta = 0.8 if args.theta is None else args.theta
    fasst = kind == "fasst"
    from dint_amd.driver import fasst_trace, tpl_trace

    wl, dtype = (wire.Workload.FASST, wire.FASST_MSG) if fasst else (wire.Workload.TPL, wire.TPL_MSG)
    from dint_amd import _lib
    piped = world == 1 and not args.force_exchange and not args.no_inputs_ready  # the `inputs_ready` leg below (not with an exchange: the routed path submits segments)
    eng = Engine(wl, n_slots=args.slots, device=dev, shard_index=rank, shard_count=world)
    rt = Router([eng], world, rank, transport=None if world > 1 else "self", n_max=BATCH) if (world > 1 or args.force_exchange) else None

    class _Server:  # 4096-request epochs through the host path (through the exchange when there are several ranks)
        def submit(self, r):
            return eng.submit(r) if rt is None else rt.submit([r])[0]

    eng.snapshot()
    n_batches = (W + K) * B
    if fasst:
        stream, recorded, cst = fasst_trace(_Server(), n_batches * BATCH, n_workers=4096, key_space=24_000_000,
                                            zipf_theta=theta if theta > 0 else None, first_worker=rank * 4096)
    else:
        stream, recorded, cst = tpl_trace(_Server(), n_batches * BATCH, n_workers=4096, key_space=24_000_000,
                                          zipf_theta=theta if theta > 0 else None, seed=0xDEADBEEF + rank)
    if rt is not None:
        rt.tighten_caps()
        rt.set_caps([rt.default_cap(BATCH)])  # the timed batches are 16 epochs long
    eng.sync()
    eng.restore()
    d_req = torch.from_numpy(np.frombuffer(stream.tobytes(), np.uint8).copy()).cuda()
    d_rep = torch.empty_like(d_req)
    msg = dtype.itemsize
    torch.cuda.synchronize()

    ahead = not args.no_ahead  # r06: every batch announces the next (dint_submit_device_ahead: its count stage rides in this batch's resolve launch)

    def run(lo, hi):  # batches [lo, hi)
        if rt is None:
            for b in range(lo, hi):
                o = b * BATCH * msg
                nxt = (d_req.data_ptr() + o + BATCH * msg, BATCH, d_rep.data_ptr() + o + BATCH * msg) if ahead and b + 1 < hi else None
                eng.submit_device(d_req.data_ptr() + o, BATCH, d_rep.data_ptr() + o, 0, ahead=nxt)
        else:
            rt.run([([d_req.data_ptr() + b * BATCH * msg], [BATCH], [d_rep.data_ptr() + b * BATCH * msg]) for b in range(lo, hi)])

    def sync():
        if rt is not None:
            rt.sync()
        eng.sync()
        torch.cuda.synchronize()

    run(0, W * B)
    sync()
    barrier(world)
    t0 = time.perf_counter()
    run(W * B, n_batches)
    sync()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, transport)
    got = d_rep.cpu().numpy().tobytes()
    # every reply byte of the recorded closed loop (one GPU: the replay applies the same requests in the same order;
    # several: a 64k batch takes 16 epochs of rank 0 before rank 1's, the recording interleaved them epoch by epoch)
    replay_ok = (got == recorded.tobytes()) if world == 1 else None
    overflow = rt.overflow() if rt is not None else 0
    # DINT_FLAG_INPUTS_READY (include/dint_abi.h): the replay hands the engine batches that are complete in HBM, in buffers of
    # their own -- the engine may then run the first half of a pass (k_lock_count, k_kv_scan_place: no table access) on a
    # stream of its own beside the previous pass's resolve kernel.  A leg, not `value`: two cross-stream dependencies per pass
    # cost most of what the overlap saves, and a lone batch gets slower (NOTEBOOK.md section 2).
    stream_ordered = None
    if piped:
        eng2 = Engine(wl, n_slots=args.slots, device=dev, flags=_lib.FLAG_INPUTS_READY)
        d_rep2 = torch.empty_like(d_rep)

        def run2(lo, hi):
            for b in range(lo, hi):
                o = b * BATCH * msg
                eng2.submit_device(d_req.data_ptr() + o, BATCH, d_rep2.data_ptr() + o, 0)

        run2(0, W * B)
        eng2.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run2(W * B, n_batches)
        eng2.sync(); torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        same2 = d_rep2.cpu().numpy().tobytes() == got  # (before the latency loop below submits batches again)
        lat2 = []
        for b in range(W * B, min(n_batches, W * B + 50)):
            eng2.sync(); torch.cuda.synchronize()
            t = time.perf_counter()
            run2(b, b + 1)
            eng2.sync(); torch.cuda.synchronize()
            lat2.append((time.perf_counter() - t) * 1e6)
        stream_ordered = {"value": round(K * B * BATCH / dt2 / 1e6, 3), "unit": "Mtxn/s", "replies_equal": same2,
                          "latency_us_p50": pct(np.array(lat2), 50),
                          "what": "the same batches through an engine created with DINT_FLAG_INPUTS_READY: count + scan / place of pass k + 1 on the engine's helper stream beside the resolve kernel of pass k"}
        del eng2, d_rep2

    lat = []
    for b in range(W * B, min(n_batches, W * B + 100)):
        sync()
        t = time.perf_counter()
        run(b, b + 1)
        sync()
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.array(lat)

    roof, extra = None, {}
    if rt is None:
        eng.timing_enable(True)
        run(W * B, min(n_batches, W * B + 200))
        sync()
        tim = eng.timing_read()
        eng.timing_enable(False)
        extra["kernels_us"] = {k: round(v["avg_us"], 3) for k, v in tim.items()}
        timed = stream[W * B * BATCH:]
        if fasst:  # 9 (req) + 9 (reply) + 8 (lock + ver read) + 8 if the op mutates the slot
            mut = float((timed["type"] != 0).mean())
            alg_bytes = BATCH * (26.0 + 8.0 * mut)
        else:      # 6 + 6 + 8 (counters read) + 8 when they are written back (a grant or a release)
            ra = recorded[W * B * BATCH:]["action"]
            mut = float(((ra == 2) | (ra == 5)).mean())
            alg_bytes = BATCH * (20.0 + 8.0 * mut)
        dom = max(tim.items(), key=lambda kv: kv[1]["avg_us"])
        # r06: with the next batch announced, the engine's "k_lock_resolve" interval is k_lock_pass -- this batch's resolve stage AND
        # the next batch's count stage in one launch: the whole pass, all its algorithmic bytes
        fused_lock = ahead and os.environ.get("DINT_LOCK_NO_FUSE") is None and dom[0] == "k_lock_resolve"
        dom_name = "k_lock_pass" if fused_lock else dom[0]
        achieved = alg_bytes / (dom[1]["avg_us"] * 1e-6) / 1e9
        # a table of a few MB lives in L2 / Infinity Cache: priced against the L2, not against HBM
        in_cache = args.slots * 8 <= 64 << 20
        peak = L2_PEAK_GBS if in_cache else HBM_PEAK_GBS
        roof = {"bound": "l2" if in_cache else "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": peak,
                "unit": "GB/s", "frac": round(achieved / peak, 5), "traffic": None,
                "alg_bytes_per_launch": int(alg_bytes), "kernel_avg_us": round(dom[1]["avg_us"], 3),
                "from_profile": profile_counters(kind, [dom_name])}
        fp = roof["from_profile"]
        if fp and fp.get("traffic_bytes"):
            roof["traffic"] = fp["traffic_bytes"]

    value = world * K * B * BATCH / dt / 1e6
    # ---- the same stream in passes of 2^20 requests: what the lock kernels do per request when a pass fills the GPU.  A 64k-
    # request pass is 64 + 768 workgroups and three dependent launches: launch-bound (BASELINE's batch size; VERDICT r04 item
    # 5).  The replies must equal those of the 64k batches -- the serial order does not depend on where a pass ends.
    big_pass = None
    if rt is None and (n_batches * BATCH) >> 20 >= 3 and not getattr(args, "no_pass_1m", False):
        big, nb_big = 1 << 20, (n_batches * BATCH) >> 20
        d_rep2 = torch.empty_like(d_rep)
        eng.restore()

        def run_big(lo, hi):
            for b in range(lo, hi):
                o = b * big * msg
                eng.submit_device(d_req.data_ptr() + o, big, d_rep2.data_ptr() + o, 0)

        run_big(0, 1)
        sync()
        t1 = time.perf_counter()
        run_big(1, nb_big)
        sync()
        dtb = time.perf_counter() - t1
        same = d_rep2[:nb_big * big * msg].cpu().numpy().tobytes() == got[:nb_big * big * msg]  # (`got`: the timed run's replies, from the empty table)
        eng.restore()
        eng.timing_enable(True)
        run_big(0, nb_big)
        sync()
        timb = eng.timing_read()
        eng.timing_enable(False)
        big_pass = {"requests_per_pass": big, "value": round((nb_big - 1) * big / dtb / 1e6, 3), "unit": "Mtxn/s",
                    "kernels_us": {k: round(v["avg_us"], 3) for k, v in timb.items()}, "replies_equal_64k_batches": same}
        del d_rep2
    if rank != 0:
        return None
    extra["pass_1m"] = big_pass
    extra["inputs_ready"] = stream_ordered
    if not args.no_rand64 and rt is None:
        rand_roofline(extra, value * 1e6, dev, 1.0, "one 8-byte slot {lock, ver} / {num_ex, num_sh} of the table per request",
                      args.slots * 8 / 2**30)
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        n_s = min(len(stream), 4_000_000 // BATCH * BATCH)  # from the empty table: checkable as a unit
        cpu = cpu_baseline_micro(kind, stream[:n_s].copy(), args.slots, got[:n_s * msg])
        if fasst and not getattr(args, "compact", False):
            extra["cpu_as_shipped"] = cpu_as_shipped_fasst(stream[:n_s].copy())
            if not args.no_shim:
                extra["shim_loopback"] = shim_loopback(stream[:n_s].copy(), dev)
    name = "lock_fasst" if fasst else "lock_2pl"
    shape = ("read / lock / validate / commit, retries on REJECT; 5-10 keys per txn" if fasst else
             "5-10 locks per txn in ascending order, exclusive with p = 0.2, REJECT releases and restarts, release in reverse")
    return {
        "metric": f"Mtxn/s ({name}: 1 txn = 1 request) + p50/p99 batch latency",
        "value": round(value, 3), "unit": "Mtxn/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"{name} on {world} MI355X: {args.slots}-slot lock table, 64k-request batches of the {name} client "
                               f"trace (4096 closed-loop workers per GPU: {shape}, "
                               f"{'Zipf-%g' % theta if theta > 0 else 'uniform'} over 24M lids); 1 step = {B} batches",
                   "batch": BATCH, "batches_per_step": B, "slots": args.slots, "parallelism": f"hash-shard x{world}", "transport": transport},
        "client": {k: cst[k] for k in ("committed", "rejects", "protocol_errors") if k in cst},
        "replay_equals_recorded": replay_ok,
        "latency_us": {"p50": pct(lat, 50), "p99": pct(lat, 99), "what": "one 64k-request batch, submit -> replies in HBM"},
        "route_overflow": overflow, "roofline": roof, "cpu_baseline": cpu, **extra,
    }


def bench_log(args, world, rank, dev, transport):
    """log_server (log_server/udp/server.cc:73-88): 53-byte COMMIT records appended to the ring, 64k-request batches
    (the reference's client resends one record forever, log_server/caladan/client.cc:153-166; here every record is
    different).  Replicas only: the log is not sharded by key, every rank appends its own stream to its own ring."""
    import torch

    from dint_amd import wire
    from dint_amd.engine import Engine

    K, W, B = args.steps, args.warmup, args.per_step
    ring = 1_000_000  # log_server/udp/utils.h:16
    eng = Engine(wire.Workload.LOG, log_entries=ring, device=dev)
    n_batches = (W + K) * B
    rng = np.random.default_rng(0x5EED + rank)
    n = n_batches * BATCH
    stream = np.zeros(n, wire.LOG_MSG)
    stream["key"] = rng.integers(0, 1 << 62, n, dtype=np.uint64)
    stream["val"][:, :8] = rng.integers(0, 256, (n, 8), dtype=np.uint8)
    stream["ver"] = rng.integers(0, 1 << 31, n)
    d_req = torch.from_numpy(np.frombuffer(stream.tobytes(), np.uint8).copy()).cuda()
    d_rep = torch.empty_like(d_req)
    msg = wire.LOG_MSG.itemsize
    torch.cuda.synchronize()

    def run(lo, hi):
        for b in range(lo, hi):
            o = b * BATCH * msg
            eng.submit_device(d_req.data_ptr() + o, BATCH, d_rep.data_ptr() + o, 0)

    def sync():
        eng.sync()
        torch.cuda.synchronize()

    run(0, W * B)
    sync()
    barrier(world)
    t0 = time.perf_counter()
    run(W * B, n_batches)
    sync()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, transport)
    got = d_rep.cpu().numpy().tobytes()
    lat = []
    for b in range(W * B, min(n_batches, W * B + 100)):
        sync()
        t = time.perf_counter()
        run(b, b + 1)
        sync()
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.array(lat)
    eng.timing_enable(True)
    run(W * B, min(n_batches, W * B + 200))
    sync()
    tim = eng.timing_read()
    eng.timing_enable(False)
    extra = {"kernels_us": {k: round(v["avg_us"], 3) for k, v in tim.items()}}
    alg_bytes = BATCH * 162.0  # 53 (req) + 53 (reply) + 56 (ring entry)
    us = sum(v["avg_us"] for v in tim.values())
    achieved = alg_bytes / (us * 1e-6) / 1e9
    roof = {"bound": "hbm", "kernel": "+".join(tim.keys()), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
            "alg_bytes_per_launch": int(alg_bytes), "kernel_avg_us": round(us, 3), "from_profile": profile_counters("log", list(tim.keys()))}
    value = world * K * B * BATCH / dt / 1e6
    # ---- the same stream in passes as large as the ring allows (1,000,000 requests): what the append kernel streams when a
    # pass fills the GPU; replies must equal those of the 64k-request batches
    big = min(ring, 1 << 20)
    nb_big = n // big
    big_pass = None
    if nb_big >= 3:
        d_rep2 = torch.empty_like(d_rep)

        def run_big(lo, hi):
            for b in range(lo, hi):
                o = b * big * msg
                eng.submit_device(d_req.data_ptr() + o, big, d_rep2.data_ptr() + o, 0)

        run_big(0, 1)
        sync()
        t1 = time.perf_counter()
        run_big(1, nb_big)
        sync()
        dtb = time.perf_counter() - t1
        same = bool(torch.equal(d_rep2[:nb_big * big * msg], d_rep[:nb_big * big * msg]))
        eng.timing_enable(True)
        run_big(0, nb_big)
        sync()
        timb = eng.timing_read()
        eng.timing_enable(False)
        usb = sum(v["avg_us"] for v in timb.values())
        achb = big * 162.0 / (usb * 1e-6) / 1e9
        big_pass = {"requests_per_pass": big, "value": round((nb_big - 1) * big / dtb / 1e6, 3), "unit": "Mtxn/s",
                    "kernel_avg_us": round(usb, 3), "achieved_GBs": round(achb, 2), "frac": round(achb / HBM_PEAK_GBS, 5),
                    "replies_equal_64k_batches": same}
        del d_rep2
    if rank != 0:
        return None
    cpu = None
    if not args.no_cpu_baseline:
        # the whole stream from the empty ring: replies and the ring's final contents must equal the oracle's
        n_s = min(n, 4_000_000 // BATCH * BATCH)
        cpu = cpu_baseline_micro("log", stream[:n_s].copy(), ring, got[:n_s * msg])
    return {
        "metric": "Mtxn/s (log_server: 1 txn = 1 appended record) + p50/p99 batch latency",
        "value": round(value, 3), "unit": "Mtxn/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"log_server on {world} MI355X: 53-byte COMMIT records into a {ring}-entry ring, 64k-request "
                               f"batches; 1 step = {B} batches", "batch": BATCH, "batches_per_step": B,
                   "parallelism": f"replicas only x{world} (the log is not sharded by key)", "transport": transport},
        "latency_us": {"p50": pct(lat, 50), "p99": pct(lat, 99), "what": "one 64k-request batch, submit -> replies in HBM"},
        "roofline": roof, "pass_1m": big_pass, "cpu_baseline": cpu, **extra,
    }


# -------------------------------------------------------------------------------------------------- store
STORE_ALG = {0: 158, 1: 162}  # READ 53 + 53 + 52, SET 53 + 53 + 12 + 44


def store_stream(n, n_sub, theta, seed, p_set=0.05):
    """store/caladan/client_udp.cc:135-147 key shape {s_id, sf_type 1..4, start_time 0/8/16}, s_id ~ Zipf(theta) over
    the populated subscribers, 95 % READ / 5 % SET (BASELINE configs[2]); SET value {end_time, 0x5a} (:56-66)"""
    from dint_amd import wire, workloads

    rng = np.random.default_rng(seed)
    z = workloads.Zipf(n_sub, theta, seed + 1)
    m = np.zeros(n, wire.STORE_MSG)
    s_id = z.sample(n).astype(np.uint64)
    m["key"] = s_id | (rng.integers(1, 5, n).astype(np.uint64) << np.uint64(32)) | (
        (rng.integers(0, 3, n) * 8).astype(np.uint64) << np.uint64(40))
    m["type"] = (rng.random(n) < p_set).astype(np.uint8)
    m["val"][:, 0] = rng.integers(0, 24, n)
    m["val"][:, 1] = 0x5A
    return m


def bench_store(args, world, rank, dev, transport):
    import torch

    from dint_amd import wire
    from dint_amd.engine import Engine
    from dint_amd.sharded import Router

    K, W, B = args.steps, args.warmup, args.per_step
    theta = 0.8 if args.theta is None else args.theta
    n_sub = args.keys // 12  # 12 rows per subscriber (store/udp/tatp.h:44-66)
    NB = 262144            # requests per batch
    eng = Engine(wire.Workload.STORE, n_rows=n_sub, device=dev, shard_index=rank, shard_count=world)
    t_setup = time.perf_counter()
    eng.populate(n_sub)
    eng.sync()
    eng.snapshot()  # (the mixes leg starts from the populated table again)
    t_setup = time.perf_counter() - t_setup
    n_batches = (K + W) * B
    stream = store_stream(NB * n_batches, n_sub, theta, 77 + rank)
    d_req = torch.from_numpy(np.frombuffer(stream.tobytes(), np.uint8).copy()).cuda()
    d_rep = torch.empty_like(d_req)
    msg = wire.STORE_MSG.itemsize
    rt = Router([eng], world, rank, n_max=NB) if world > 1 else None
    torch.cuda.synchronize()
    # r06: the replay answers in place (store/udp/server.cc:75-97 mutates the received struct) and announces the next batch
    inplace, ahead = args.replay == "inplace" and rt is None, not args.no_ahead and rt is None

    def reset():  # the receive buffers hold the pristine requests again (outside every timed region: the NIC's DMA)
        if inplace:
            d_rep.copy_(d_req)
            torch.cuda.synchronize()

    def run(lo, hi):
        src = d_rep if inplace else d_req
        for b in range(lo, hi):
            o = b * NB * msg
            if rt is None:
                nxt = (src.data_ptr() + o + NB * msg, NB, d_rep.data_ptr() + o + NB * msg) if ahead and b + 1 < hi else None
                eng.submit_device(src.data_ptr() + o, NB, d_rep.data_ptr() + o, 0, ahead=nxt)
            else:
                rt.step([d_req.data_ptr() + o], [NB], [d_rep.data_ptr() + o])

    def sync():
        if rt is not None:
            rt.sync()
        eng.sync()
        torch.cuda.synchronize()

    reset()
    run(0, W * B)
    sync()
    barrier(world)
    t0 = time.perf_counter()
    run(W * B, n_batches)
    sync()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, transport)
    got = d_rep.cpu().numpy()
    reset()
    lat = []
    for b in range(W * B, min(n_batches, W * B + 100)):
        sync()
        t = time.perf_counter()
        run(b, b + 1)
        sync()
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.array(lat)
    roof, extra, cpu = None, {}, None
    ty = stream[W * B * NB:]["type"]
    if rt is None:
        reset()
        eng.timing_enable(True)
        run(W * B, min(n_batches, W * B + 200))
        sync()
        tim = eng.timing_read()
        eng.timing_enable(False)
        extra["kernels_us"] = {k: round(v["avg_us"], 3) for k, v in tim.items()}
        alg_all = NB * (STORE_ALG[0] * float((ty == 0).mean()) + STORE_ALG[1] * float((ty == 1).mean()))
        # the table requests are answered by k_kv_resolve and -- the hot keys -- by k_kv_big behind it: priced together, as
        # for tatp / smallbank; the 53 request bytes of every request are k_kv_part's, which reads them (bench_txn)
        part_b, alg = NB * float(msg), alg_all - NB * float(msg)
        t_part = tim.get("k_kv_part", {"avg_us": 0.0})["avg_us"]
        z = {"avg_us": 0.0}
        us = (tim.get("k_kv_resolve", z)["avg_us"] + tim.get("k_kv_pass", z)["avg_us"] + tim.get("k_kv_hot", z)["avg_us"] + tim.get("k_kv_big", z)["avg_us"] +
              tim.get("k_kv_late", z)["avg_us"])
        ach = alg / (us * 1e-6) / 1e9
        one_launch = os.environ.get("DINT_KV_NO_FUSE", "0") in ("", "0")
        if one_launch:  # r06: k_kv_pass (resolve + hot-key workers + the next batch's partition) + k_kv_late: the whole pass is the priced stage
            alg, us = alg_all, us + t_part
            ach = alg / (us * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": "k_kv_pass+k_kv_late" if one_launch else "k_kv_resolve+k_kv_hot+k_kv_big", "achieved": round(ach, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "alg_bytes_per_launch": int(alg),
                "kernel_avg_us": round(us, 3),
                "from_profile": profile_counters("store", (["k_kv_pass", "k_kv_late"] + ([] if ahead else ["k_kv_part"])) if one_launch else
                                                 ["k_kv_resolve", "k_kv_hot", "k_kv_hot_part", "k_kv_big"])}
        fp = roof["from_profile"]
        if fp and fp.get("traffic_bytes"):
            roof["traffic"] = fp["traffic_bytes"]
            roof["traffic_over_alg"] = round(fp["traffic_bytes"] / max(1.0, alg), 3)
        roof["kernels"] = [{"kernel": k, "kernel_avg_us": round(tim[k]["avg_us"], 3)} for k in tim]
        roof["kernels"][0].update({"alg_bytes_per_launch": int(part_b), "frac": round(part_b / max(t_part, 1e-9) / 1e3 / HBM_PEAK_GBS, 5)})
        chain_us = us if one_launch else t_part + us
        roof["pass"] = {"alg_bytes": int(alg_all), "chain_us": round(chain_us, 3),
                        "frac": round(alg_all / max(chain_us, 1e-9) / 1e3 / HBM_PEAK_GBS, 5),
                        "what": "the pass: algorithmic bytes of all its requests / (part + resolve + big)"}
        roof["gpu"] = {"alg_bytes_per_request": round(alg_all / NB, 1), "achieved": round(K * B * NB / dt * alg_all / NB / 1e9, 2),
                       "frac": round(K * B * NB / dt * alg_all / NB / 1e9 / HBM_PEAK_GBS, 5),
                       "what": "requests/s of the timed region x mean algorithmic bytes per request / 8 TB/s"}
    value = world * K * B * NB / dt / 1e6
    mixes = {}
    if rank != 0:
        return None
    if not args.no_rand64 and rt is None:
        rt_ = np.frombuffer(got.tobytes(), wire.STORE_MSG)[W * B * NB:]["type"]
        hit = float(((rt_ == 3) | (rt_ == 5)).mean())  # GRANT_READ / SET_ACK: the row's value sector(s) are touched
        rand_roofline(extra, value * 1e6, dev, 1.0 + 1.5 * hit,
                      "bucket header sector + 1.5 value sectors for a request that finds its row (40-byte values at 64 + 40 k "
                      "straddle a sector boundary for two of the four slots)", eng_table_gb(eng))
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle as orc

        o = orc.StoreOracle(n_sub * 18 // 4, n_sub)
        n_s = min(len(stream), 8 * NB)
        t = time.perf_counter()
        rep = o.replay(stream[:n_s].copy())
        dtc = time.perf_counter() - t
        cpu = {"value": round(n_s / dtc / 1e6, 4), "unit": "Mtxn/s", "cores": 1, "kind": "port", "host_cpu": host_cpu(),
               "sample": f"the first {n_s} requests of the bench stream, oracle/dint_oracle.c ({n_sub * 12} keys), 1 thread",
               "oracle_parity": {"requests": n_s, "ok": rep.tobytes() == got.tobytes()[:n_s * msg]}}
    # ---- the reference client's other two mixes (store/caladan/client_udp.cc:56-66: `parallel` = 100 % READ, `contention` =
    # 50 % READ / 50 % SET), same table size and key distribution, 16 batches each.  From a known state -- the populated
    # table + the first n_s requests of the bench stream, where the oracle above stands -- so every reply is checked.
    if rt is None and not args.no_mixes:
        n_s = min(len(stream), 8 * NB)
        eng.restore()
        eng.submit_device(d_req.data_ptr(), n_s, d_rep.data_ptr(), 0)  # (several passes: they look ahead at each other)
        sync()
        for name, p_set in (("100/0", 0.0), ("50/50", 0.5)):
            ms = store_stream(NB * 16, n_sub, theta, 991 + rank, p_set)
            dq = torch.from_numpy(np.frombuffer(ms.tobytes(), np.uint8).copy()).cuda()
            dp = torch.empty_like(dq)
            sync()
            tm = time.perf_counter()
            for b in range(16):
                eng.submit_device(dq.data_ptr() + b * NB * msg, NB, dp.data_ptr() + b * NB * msg, 0)
            sync()
            tm = time.perf_counter() - tm
            mixes[name] = {"value": round(16 * NB / tm / 1e6, 3), "unit": "Mtxn/s", "read_write": name, "batches": 16}
            if cpu is not None:
                mixes[name]["oracle_parity"] = {"requests": len(ms), "ok": o.replay(ms.copy()).tobytes() == dp.cpu().numpy().tobytes()}
            del dq, dp
    return {
        "metric": "Mtxn/s (store: 1 txn = 1 request) + p50/p99 batch latency",
        "value": round(value, 3), "unit": "Mtxn/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "mixes": mixes or None,
        "config": {"workload": f"store KV on {world} MI355X: {n_sub * 12} keys x 40-B values, 95/5 read/write, "
                               f"s_id ~ Zipf-{theta}, {NB}-request batches; 1 step = {B} batches", "keys": n_sub * 12, "batch": NB,
                   "batches_per_step": B, "parallelism": f"hash-shard x{world}", "transport": transport,
                   "replay": "in place" if inplace else "separate reply buffers", "look_ahead": bool(ahead)},
        "latency_us": {"p50": pct(lat, 50), "p99": pct(lat, 99), "what": "one batch, submit -> replies in HBM"},
        "roofline": roof, "cpu_baseline": cpu, "setup_s": round(t_setup, 2), **extra,
    }


def eng_table_gb(eng, n_engines=1):
    """HBM footprint of the kv tables of `n_engines` engines like `eng`, in GB (for the roofline's table size)"""
    per = 256 if eng.msg_size != 23 else 128
    buckets = sum(eng.hash_size(t) for t in range({53: 1, 55: 5, 23: 2}[eng.msg_size]))
    return n_engines * buckets * 1.25 * per / 2**30


# ----------------------------------------------------------------------------------- tatp / smallbank
# algorithmic bytes per request: request + reply + the row / lock bytes the op needs
TATP_ALG = {0: 162, 1: 126, 2: 126, 12: 170, 18: 170, 22: 170, 13: 162, 19: 162, 23: 162, 14: 174, 24: 174}
TATP_LOG_TYPES = (14, 24)
SB_ALG = {0: 78, 1: 78, 2: 58, 3: 58, 4: 66, 5: 66, 6: 78}  # 23 + 23 + {20 row + 12 counters | 12 | 20 | 32 log}
SB_LOG_TYPES = (6,)


def epoch_host(rp, e, shards=(0, 1, 2)):
    """host copies (requests, recorded replies) of epoch e's batches, pulled from HBM on demand"""
    from dint_amd import wire

    dt = {55: wire.TATP_MSG, 23: wire.SB_MSG}[rp.msg]
    req = [np.frombuffer(rp.d_req[e][s].cpu().numpy().tobytes(), dt) if s in shards else None for s in range(3)]
    rep = [np.frombuffer(rp.d_want[e][s].cpu().numpy().tobytes(), dt) if s in shards else None for s in range(3)]
    return req, rep


def cpu_baseline_txn(kind, rp, done, n_rows, lo, hi):
    """Shard server 0's recorded request stream of epochs [0, hi) replayed on one host core by the CPU port
    (oracle/dint_oracle.c); epochs [lo, hi) are timed.  Every reply must equal what the GPU engine answered when the
    stream was recorded (oracle parity on the exact bench stream).  A transaction costs `ops_per_txn` requests over
    the three servers, so one core serving all three streams back to back completes (ops/s) / ops_per_txn
    transactions per second."""
    from oracle import oracle as orc

    o = orc.TatpOracle(n_rows) if kind == "tatp" else orc.SmallbankOracle(n_rows)
    ok, checked = True, 0
    n, dt = 0, 0.0
    for e in range(hi):
        rq, want = epoch_host(rp, e, (0,))
        req = rq[0]
        t = time.perf_counter()
        rep = o.replay(req)
        if e >= lo:
            dt += time.perf_counter() - t
            n += len(req)
        ok = ok and rep.tobytes() == want[0].tobytes()
        checked += len(req)
    ops_all = rp.ops(lo, hi)
    ops_per_txn = ops_all / max(1, sum(done[lo:hi]))
    return {"value": round(n / dt / ops_per_txn / 1e6, 4), "unit": "Mtxn/s", "cores": 1, "kind": "port",
            "ops_per_s": round(n / dt), "ops_per_txn": round(ops_per_txn, 3), "host_cpu": host_cpu(),
            "sample": f"{n} requests = shard server 0's stream of {hi - lo} bench epochs, oracle/dint_oracle.c "
                      f"({n_rows} rows), 1 thread",
            "oracle_parity": {"epochs": hi, "requests": checked, "ok": ok,
                              "what": "GPU replies of shard server 0 vs the CPU oracle, every byte, on the bench stream"}}


REF_TATP_SUBSCRIBERS = 7_000_000  # tatp/udp/tatp.h:28 (constexpr: the unmodified server has no other size)


def cpu_reference_tatp(ref, args, dev, C, zipf, n_epochs=12, shipped=None, extra=None):
    """BASELINE.md 3(1) for the headline workload: the UNMODIFIED tatp/udp/server_shard.cc (oracle/_ref/ref_tatp,
    sockets interposed, 1 thread) against the engines on the configuration the reference is compiled for -- 7M
    subscribers on both sides.  A second shard group is populated at 7M rows, the closed loop runs `n_epochs` epochs
    through it, shard server 0's request stream is replayed by the reference and every reply byte is compared (bytes
    the reference's populate leaves unassigned are masked on rows never written: oracle.mask_populate_garbage)."""
    import torch

    from dint_amd import wire
    from dint_amd.driver import Driver
    from dint_amd.replay import Replay, ShardGroup, record
    from oracle import oracle as orc

    n_rows = REF_TATP_SUBSCRIBERS
    t_setup = time.perf_counter()
    grp = ShardGroup(wire.Workload.TATP, n_rows, device=dev, rank=0, world=1, transport="self", n_max=KV_PASS)
    grp.sync()
    grp.snapshot()
    drv = Driver(wire.Workload.TATP, C, n_rows, first_client=0, zipf_theta=zipf)
    trace, done = record(drv, grp, n_epochs)
    grp.sync()
    rp = Replay(trace, grp.msg)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    gpu = []
    for _ in range(3):
        grp.restore()
        grp.sync()
        t = time.perf_counter()
        rp.run(grp, 0, n_epochs)
        grp.sync()
        gpu.append(sum(done) / (time.perf_counter() - t) / 1e6)
    rp.check(0, n_epochs)
    req = np.concatenate([trace[e][0][0] for e in range(n_epochs)])
    want = np.concatenate([trace[e][1][0] for e in range(n_epochs)])
    ops_per_txn = sum(sum(len(r) for r in trace[e][0]) for e in range(n_epochs)) / max(1, sum(done))
    if shipped is not None and extra is not None:
        extra["cpu_as_shipped_tatp"] = cpu_as_shipped_tatp(shipped, trace, ops_per_txn, args)
    del grp, rp
    if not ref.wait_populated(args.ref_timeout):
        return {"error": f"reference server not populated within {args.ref_timeout:.0f} s of the GPU legs ending"}
    rep, st = ref.replay(req)
    a = orc.mask_populate_garbage("tatp", rep).tobytes()
    b = orc.mask_populate_garbage("tatp", want).tobytes()
    return {"value": round(st["ops_per_s"] / ops_per_txn / 1e6, 4), "unit": "Mtxn/s", "cores": 1, "kind": "reference",
            "ops_per_s": round(st["ops_per_s"]), "ops_per_txn": round(ops_per_txn, 3), "host_cpu": host_cpu(),
            "sample": f"{len(req)} requests = shard server 0's stream of the first {n_epochs} closed-loop epochs at "
                      f"{n_rows} subscribers, unmodified tatp/udp/server_shard.cc (oracle/_ref/ref_tatp, sockets "
                      f"interposed in-process, 1 thread, populate {ref.populate_s:.0f} s not timed)",
            "gpu_same_config": {"value": round(max(gpu), 3), "unit": "Mtxn/s", "runs": [round(g, 1) for g in gpu],
                                "subscribers": n_rows, "epochs": n_epochs, "setup_s": round(t_setup, 1),
                                "what": "the same epochs replayed from HBM through the three engines at 7M subscribers"},
            "reference_parity": {"requests": len(req), "ok": a == b,
                                 "what": "GPU replies of shard server 0 vs the unmodified reference server, every byte"}}


def host_path(grp, rp, lo, hi):
    """The boundary the reference's servers sit behind hands over HOST buffers: the same recorded batches through
    dint_submit_async / dint_wait from page-locked memory (H2D + kernels + D2H, three staging slots per engine).
    Returns per-epoch latency (submit of the three batches -> all replies in host memory) and the pipelined rate."""
    from dint_amd.engine import Pinned

    msg = grp.msg
    bufs, want = [], []
    for e in range(lo, hi):
        rq, rw = epoch_host(rp, e)
        row = []
        for s in range(3):
            b = rq[s].tobytes()
            pin, pout = Pinned(max(len(b), 1)), Pinned(max(len(b), 1))
            pin.array[:len(b)] = np.frombuffer(b, np.uint8)
            row.append((pin, pout, len(b) // msg))
        bufs.append(row)
        want.append([rw[s].tobytes() for s in range(3)])
    grp.sync()
    lat = []
    for row in bufs:  # one epoch at a time: latency
        t = time.perf_counter()
        tk = [grp.engines[s].submit_async(row[s][0].ptr, row[s][2], row[s][1].ptr) for s in range(3)]
        for s in range(3):
            grp.engines[s].wait(tk[s])
        lat.append((time.perf_counter() - t) * 1e6)
    ok = all(row[s][1].array[:row[s][2] * msg].tobytes() == want[i][s] for i, row in enumerate(bufs) for s in range(3))
    return np.array(lat), bufs, ok


def host_path_rate(grp, bufs):
    """all epochs submitted back to back (the engines pipeline copies and kernels), one wait at the end"""
    grp.sync()
    t = time.perf_counter()
    tk = [0, 0, 0]
    for row in bufs:
        for s in range(3):
            tk[s] = grp.engines[s].submit_async(row[s][0].ptr, row[s][2], row[s][1].ptr)
    for s in range(3):
        grp.engines[s].wait(tk[s])
    return time.perf_counter() - t


def type_histogram(rp, lo, hi, type_off):
    """request types of epochs [lo, hi) (all three servers), counted on the device"""
    import torch

    h = torch.zeros(256, dtype=torch.long, device="cuda")
    launches = 0
    for e in range(lo, hi):
        for s in range(3):
            if rp.counts[e][s]:
                h += torch.bincount(rp.d_req[e][s][type_off::rp.msg].long(), minlength=256)
                launches += -(-rp.counts[e][s] // KV_PASS)
    return h.cpu().numpy(), launches


def txn_U(kind, rp, lo, hi):
    """random 64-byte table sectors a request touches, by the layout (dint_kv_core.h), averaged over the requests of
    epochs [lo, hi): tatp -- the bucket's header sector for every table request, + 1.5 value sectors when a row is read or
    written (READ that finds it, COMMIT_*, INSERT_*; 40-byte values straddle a sector for two slots of four), log
    appends are sequential (0); smallbank -- header sector + the sector holding the four 8-byte values and the
    counters for every table request."""
    import torch

    tot, units = 0, 0.0
    for e in range(lo, hi):
        for s in range(3):
            n = rp.counts[e][s]
            if not n:
                continue
            rq = rp.d_req[e][s][1::rp.msg].long()
            rep = rp.d_want[e][s][1::rp.msg].long()
            tot += n
            if kind == "tatp":
                is_log = (rq == 14) | (rq == 24)
                row = (rep == 4) | (rq == 12) | (rq == 13) | (rq == 18) | (rq == 19)
                units += float((~is_log).sum() + 1.5 * row.sum())
            else:
                units += 2.0 * float((rq != 6).sum())
    return units / max(1, tot)


def bench_txn(args, world, rank, dev, transport, kind):
    import torch

    from dint_amd import wire
    from dint_amd.driver import Driver
    from dint_amd.replay import Replay, ShardGroup

    K, W, B = args.steps, args.warmup, args.per_step
    C = args.clients
    if kind == "tatp":
        # weak scaling: every GPU brings its clients AND its subscribers (--subscribers is per GPU), so the contention per
        # key -- and the load of the hot key's home rank -- does not grow with N (4.2M lock-step clients on
        # 1M subscribers at 8 GPUs would be neither weak scaling nor BASELINE's configuration)
        wl, n_rows, theta = wire.Workload.TATP, args.subscribers * world, (0.8 if args.theta is None else args.theta)
        alg_tab, log_types, dtype = TATP_ALG, TATP_LOG_TYPES, "u64"
    else:
        wl, theta = wire.Workload.SMALLBANK, (0.99 if args.theta is None else args.theta)
        n_rows = args.accounts if args.accounts else 10_000_000 * world
        alg_tab, log_types, dtype = SB_ALG, SB_LOG_TYPES, "u64"
    zipf = theta if theta > 0 else None
    E0, E1 = W * B, (W + K) * B  # timed epochs
    t_setup = time.perf_counter()
    grp = ShardGroup(wl, n_rows, device=dev, rank=rank, world=world, transport=None if world > 1 else "self",
                     force_exchange=args.force_exchange, n_max=KV_PASS, flags=int(os.environ.get("DINT_BENCH_FLAGS", "0")))
    grp.sync()
    grp.snapshot()
    drv = Driver(wl, C, n_rows, first_client=rank * C, zipf_theta=zipf)
    # the closed loop, once, through the real engines; every epoch stays in HBM.  The replay answers in place and announces
    # every next batch (r06; --replay copy --no-ahead = r05's replay); through the exchange the batches are routed copies anyway
    inplace = args.replay == "inplace" and grp.router is None
    rp, done, _ = Replay.recording(drv, grp, E1, inplace=inplace, ahead=not args.no_ahead and grp.router is None)
    stats = drv.stats()
    grp.sync()
    caps = grp.router.tighten_caps() if grp.router is not None else None  # slot capacities from the recorded maxima
    grp.restore()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    # The recording above is host-bound (seconds of a mostly idle GPU): the first GPU-bound stretch after it runs at idle clocks
    # for tens of milliseconds (r05: the W warm-up steps alone -- 7 ms -- left the timed region 3x slower than its repeats).
    # Untimed spin-up first: the warm-up epochs replayed until the GPU has been busy for ~0.3 s; then the state of the
    # recording's start again, the W warm-up steps, the K timed steps.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        rp.reset(0, max(1, E0))
        rp.run(grp, 0, max(1, E0))
        grp.sync()
        grp.restore()
    grp.sync()
    rp.reset()  # (in-place replay: the receive buffers hold the pristine requests again -- before the clock starts)
    rp.run(grp, 0, E0)
    grp.sync()
    barrier(world)
    t0 = time.perf_counter()
    rp.run(grp, E0, E1)
    t_issue = time.perf_counter() - t0  # host time to enqueue the K steps (the GPU runs behind it)
    grp.sync()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, transport)
    rp.check(0, E1)  # parity with the recorded closed-loop run, every reply byte
    overflow = grp.router.overflow() if grp.router is not None else 0
    ref = shipped = None
    compact = getattr(args, "compact", False)
    if kind == "tatp" and world == 1 and not (args.no_cpu_baseline or args.no_cpu_reference or compact):
        from oracle import oracle as orc  # cpu_baseline leg only: the reference populates (minutes, one host core)
        if orc.ref_available("tatp"):     # while the remaining GPU legs run; the headline region above is over
            ref = orc.RefServer("tatp")
        if orc.loopback_tatp_available() and not args.no_as_shipped:  # ... and the as-shipped deployment: three servers
            try:
                shipped = orc.TatpAsShipped(server_threads=8)  # `server_shard <id> 8`, exp/run_tatp.sh
            except Exception:
                shipped = None

    txns = sum_over_ranks(sum(done[E0:E1]), world, transport)
    ops = sum_over_ranks(rp.ops(E0, E1), world, transport)
    value = txns / dt / 1e6

    # run-to-run spread of the same K steps (state restored before each)
    repeats = []
    for _ in range(0 if compact else 4):
        grp.restore()
        rp.reset()
        rp.run(grp, 0, E0)
        grp.sync()
        barrier(world)
        t1 = time.perf_counter()
        rp.run(grp, E0, E1)
        grp.sync()
        barrier(world)
        repeats.append(txns / max_over_ranks(time.perf_counter() - t1, world, transport) / 1e6)

    # per-epoch latency, device side: submit of the three batches -> all replies visible in HBM
    grp.restore()
    rp.reset()
    lat = []
    for e in range(min(E1, 120)):
        grp.sync()
        t = time.perf_counter()
        rp.run(grp, e, e + 1)
        grp.sync()
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.array(lat[min(E0, len(lat) // 2):])

    roof, extra = None, {}
    if world == 1 and grp.router is None:
        grp.restore()
        rp.reset()
        for e in grp.engines:
            e.timing_enable(True)
        n_t = min(E1, 200)
        st0 = [e.stats() for e in grp.engines]
        big0 = sum(x["big_bin_requests"] for x in st0)
        rp.run(grp, 0, n_t)
        grp.sync()
        tims = [e.timing_read() for e in grp.engines]
        st1 = [e.stats() for e in grp.engines]
        big_req = sum(x["big_bin_requests"] for x in st1) - big0
        # what no closed form of k_kv_hot covered (the slow passes): requests, and work items by kind {sub as listed, solo, pieces}
        extra["late"] = {"requests": sum(b["late_requests"] - a["late_requests"] for a, b in zip(st0, st1)),
                         "items_by_kind": [sum(b["late_items"][k] - a["late_items"][k] for a, b in zip(st0, st1)) for k in range(3)],
                         "passes": 3 * n_t}
        for e in grp.engines:
            e.timing_enable(False)
        names = list(tims[0].keys())
        avg = {k: float(np.mean([t[k]["avg_us"] for t in tims])) for k in names}
        extra["kernels_us"] = {k: round(v, 3) for k, v in avg.items()}
        # A pass = k_kv_part (every request is read, classified, hashed and put into its coarse bin; log requests are finished
        # there) -> the RESOLVE STAGE = k_kv_resolve (every coarse bin: the table requests but those of the hot keys) +
        # k_kv_big (the hot keys), two launches that answer the table requests between them.  Every kv workload is priced
        # the same way (store always was): `roofline` = the resolve stage, both kernels' time against the bytes of the
        # table requests -- it is the longest piece of the chain on every box, so the line's meaning does not move with a
        # few microseconds (r04 picked "the longer kernel" and flipped between 0.064 and 0.004).
        # Algorithmic bytes are credited to the kernel that must move them: the 55 / 23 request bytes of
        # EVERY request to k_kv_part, which reads them (+ reply and ring record of the log requests it finishes); the reply
        # and the row / lock bytes of a table request to the resolve stage.
        hist, launches = type_histogram(rp, 0, n_t, 1)
        msgb = rp.msg
        n_all = sum(int(hist[c]) for c in alg_tab)
        n_tab = sum(int(hist[c]) for c in alg_tab if c not in log_types)
        tab_b = sum((b - msgb) * int(hist[c]) for c, b in alg_tab.items() if c not in log_types)  # reply + row / lock bytes
        part_b = msgb * n_all + sum((b - msgb) * int(hist[c]) for c, b in alg_tab.items() if c in log_types)
        f_big = big_req / max(1, n_tab)
        L = max(1, launches)
        # (the engine's timer: {k_kv_part, k_kv_pass, k_kv_late} when a pass is one launch -- store / tatp, r06 --, else r05's four)
        t_part, t_res = avg.get("k_kv_part", 0.0), avg.get("k_kv_resolve", 0.0) + avg.get("k_kv_pass", 0.0)
        t_big = avg.get("k_kv_hot", 0.0) + avg.get("k_kv_big", 0.0) + avg.get("k_kv_late", 0.0)  # the hot keys: closed forms (k_kv_hot), then what they do not cover

        def priced(name, us, nbytes):
            ach = nbytes / L / max(us, 1e-9) / 1e3  # bytes per launch / us -> GB/s
            return {"kernel": name, "kernel_avg_us": round(us, 3), "alg_bytes_per_launch": int(nbytes / L),
                    "achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5)}

        one_launch = kind != "smallbank" and os.environ.get("DINT_KV_NO_FUSE", "0") in ("", "0")
        if one_launch:
            # r06: a store / tatp pass is ONE launch, k_kv_pass -- the resolve stage, the hot-key workers and (look-ahead replay) the
            # NEXT batch's partition -- plus k_kv_late (what no closed form covered; usually an empty launch).  The engine's
            # "k_kv_resolve" interval is k_kv_pass, "k_kv_big" is k_kv_late, "k_kv_part" the first pass's partition only (or, without
            # look-ahead, every pass's).  The priced stage is the WHOLE pass: every algorithmic byte over the whole chain.
            stage = priced("k_kv_pass+k_kv_late" + ("" if rp.ahead else "+k_kv_part"), t_part + t_res + t_big, tab_b + part_b)
        elif rp.ahead and kind == "smallbank":
            # smallbank, look-ahead replay: k_kv_pass (this pass's resolve workgroups + the next batch's partition) -> k_kv_big (the big
            # subs; a hot account's row in pieces, kv_sb_item).  The whole pass is the priced stage.
            stage = priced("k_kv_pass+k_kv_big", t_part + t_res + t_big, tab_b + part_b)
        elif rp.ahead:
            # look-ahead replay, DINT_KV_NO_FUSE: k_kv_resolve -> k_kv_hot_part (the hot keys AND the next batch's partition) -> k_kv_late
            stage = priced("k_kv_resolve+k_kv_hot_part+k_kv_big", t_part + t_res + t_big, tab_b + part_b)
        else:
            stage = priced("k_kv_resolve+k_kv_hot+k_kv_big", t_res + t_big, tab_b)
        # `traffic` is not measured inside this run (rocprofv3 cannot attach to itself): null here; the PMC figures of
        # the same command live in profiles/ and are quoted under from_profile only for the same kernel sources
        roof = {"bound": "hbm", **stage, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "requests_in_big_bins": round(f_big, 4),
                # (the kernels of THIS configuration's pass only: the profile also holds the recording's kernels under their own names)
                "from_profile": profile_counters(kind, (["k_kv_pass", "k_kv_late"] + ([] if rp.ahead else ["k_kv_part"])) if one_launch else
                                                 (["k_kv_pass", "k_kv_big"] if kind == "smallbank" and rp.ahead else
                                                  ["k_kv_resolve", "k_kv_hot_part", "k_kv_big"] if rp.ahead else ["k_kv_resolve", "k_kv_hot", "k_kv_big"]))}
        fp = roof["from_profile"]
        if fp and fp.get("traffic_bytes"):
            roof["traffic"] = fp["traffic_bytes"]
            roof["traffic_over_alg"] = round(fp["traffic_bytes"] / max(1.0, stage["alg_bytes_per_launch"]), 3)
        if one_launch:
            roof["kernels"] = [priced("k_kv_part" + (" (first pass only)" if rp.ahead else ""), t_part, 0 if rp.ahead else part_b),
                               priced("k_kv_pass", t_res, tab_b + (part_b if rp.ahead else 0)),
                               priced("k_kv_late", t_big, 0)]
        elif rp.ahead:
            roof["kernels"] = [priced("k_kv_part (first pass only)", t_part, 0), priced("k_kv_resolve", t_res, tab_b * (1.0 - f_big)),
                               priced("k_kv_hot_part+k_kv_big", t_big, tab_b * f_big + part_b)]
        else:
            roof["kernels"] = [priced("k_kv_part", t_part, part_b), priced("k_kv_resolve", t_res, tab_b * (1.0 - f_big)),
                               priced("k_kv_hot+k_kv_big", t_big, tab_b * f_big)]
        # the whole pass of one engine: all its algorithmic bytes over the serial chain of its three kernels (events on the
        # engine's stream, launch gaps included) -- what bounds a step, which is one engine's chain
        chain = t_part + t_res + t_big
        roof["pass"] = {"alg_bytes": int((part_b + tab_b) / L), "chain_us": round(chain, 3),
                        "frac": round((part_b + tab_b) / L / max(chain, 1e-9) / 1e3 / HBM_PEAK_GBS, 5),
                        "what": "one engine's pass: algorithmic bytes of all its requests / (part + resolve + big)"}
        # ... and the GPU as a whole in the timed region: three engines' chains side by side
        alg_all = sum(b * int(hist[c]) for c, b in alg_tab.items()) / max(1, n_all)  # mean algorithmic bytes per request
        roof["gpu"] = {"alg_bytes_per_request": round(alg_all, 1), "achieved": round(ops / dt * alg_all / 1e9, 2),
                       "frac": round(ops / dt * alg_all / 1e9 / HBM_PEAK_GBS, 5),
                       "what": "requests/s of the timed region x mean algorithmic bytes per request / 8 TB/s"}
        # (scalar copies: the driver's record of the line keeps the scalars of `roofline`, not its nested objects)
        roof["pass_chain_us"], roof["pass_frac"] = roof["pass"]["chain_us"], roof["pass"]["frac"]
        roof["gpu_achieved"], roof["gpu_frac"] = roof["gpu"]["achieved"], roof["gpu"]["frac"]

    # ---- the closed loop itself, resident on the GPU: the same clients as device code emit the same
    # stream (tests/test_gpu_gdriver.py), the engines read the batch sizes on the device, nothing crosses PCIe
    closed = None
    if world == 1 and grp.router is None and not args.no_closed_loop and not compact:
        from dint_amd.driver import GpuDriver
        from dint_amd.replay import GpuLoop

        grp.restore()
        cap = min(min(e.pass_max for e in grp.engines), int(1.25 * max(max(c) for c in rp.counts)) + 4096)
        gd = GpuDriver(wl, C, n_rows, cap, first_client=rank * C, zipf_theta=zipf)
        loop = GpuLoop(grp, gd)
        loop.epochs(E0)
        loop.sync()
        tx0 = gd.stats()["txns"]
        t1 = time.perf_counter()
        loop.epochs(E1 - E0)
        loop.sync()
        dtc = time.perf_counter() - t1
        gs = gd.stats()
        # same seeds, same servers => the device clients must have finished exactly the transactions the host
        # driver finished while the trace was recorded
        same = all(gs[k] == stats[k] for k in ("txns", "committed", "by_type", "committed_by_type"))
        closed = {"value": round((gs["txns"] - tx0) / dtc / 1e6, 3), "unit": "Mtxn/s", "ms_per_epoch": round(dtc / (E1 - E0) * 1e3, 5),
                  "epochs": E1 - E0, "equals_host_driver_run": bool(same), "overflow": gs["overflow"],
                  "what": "GPU-resident clients (k_txn_emit, the replies consumed by the next emit) + the three shard servers "
                          "(one launch set per epoch on the clients' stream: dint_submit_segments_multi), closed loop, no host round trip"}
        del loop, gd
        # ... and with the clients in two groups that take turns at the servers (GpuLoop): one group's consume / emit
        # kernels run while the servers answer the other's batch.  Checked against a host run of two Drivers taking
        # turns the same way for a few epochs; then timed from a restored state.
        from dint_amd.driver import Driver
        G, Cg, n_chk = 2, C // 2, 6
        grp.restore()
        hosts = [Driver(wl, Cg, n_rows, first_client=rank * C + g * Cg, zipf_theta=zipf) for g in range(G)]
        for _ in range(n_chk):
            for h in hosts:
                h.consume(grp.submit(h.next()))
        grp.restore()
        gds = [GpuDriver(wl, Cg, n_rows, cap, first_client=rank * C + g * Cg, zipf_theta=zipf) for g in range(G)]
        loop = GpuLoop(grp, gds)
        loop.epochs(n_chk)
        loop.sync()
        same2 = all(g.stats()[k] == h.stats()[k] for g, h in zip(gds, hosts) for k in ("txns", "committed", "by_type", "committed_by_type"))
        loop.epochs(E0)
        loop.sync()
        tx0 = sum(g.stats()["txns"] for g in gds)
        t1 = time.perf_counter()
        loop.epochs(E1 - E0)
        loop.sync()
        dtc = time.perf_counter() - t1
        closed["two_groups"] = {"value": round((sum(g.stats()["txns"] for g in gds) - tx0) / dtc / 1e6, 3), "unit": "Mtxn/s",
                                "ms_per_round": round(dtc / (E1 - E0) * 1e3, 5), "clients_per_group": Cg,
                                "equals_host_run_of_two_drivers": bool(same2), "checked_epochs": n_chk,
                                "overflow": sum(g.stats()["overflow"] for g in gds),
                                "what": "the same clients in two groups taking turns at the servers, each group on its own stream (slower: the "
                                        "two groups' kernels wait on tickets / look-backs while they share the GPU)"}
        del loop, gds, hosts

    host = {}
    if world == 1 and grp.router is None and not args.no_host_path and not compact:
        grp.restore()
        n_h = min(E1, 24)
        hl, bufs, ok = host_path(grp, rp, 0, n_h)
        grp.restore()
        dth = host_path_rate(grp, bufs)
        tx_h, ops_h = sum(done[:n_h]), rp.ops(0, n_h)
        host = {"latency_host_us": {"p50": pct(hl[n_h // 4:], 50), "p99": pct(hl[n_h // 4:], 99),
                                    "what": "dint_submit_async x3 + dint_wait from page-locked host buffers: H2D + kernels + D2H"},
                "value_pcie": round(tx_h / dth / 1e6, 3), "Mops_s_pcie": round(ops_h / dth / 1e6, 3),
                "pcie_parity_ok": bool(ok)}
        del bufs
    if rank != 0:
        return None
    if not args.no_rand64 and world == 1 and grp.router is None:
        u_what = ("bucket header sector per table request + 1.5 value sectors when a row is read or written; log appends are sequential"
                  if kind == "tatp" else "bucket header sector + the sector of the four values and the counters, per table request")
        rand_roofline(extra, ops / dt, dev, txn_U(kind, rp, E0, min(E1, E0 + 32)), u_what, eng_table_gb(grp.engines[0], 3))
        r64 = extra.get("roofline_rand64")
        if roof is not None and r64:  # the north star's own fraction, where the driver's record keeps it
            roof["rand64_frac"], roof["rand64_U"], roof["rand64_gathers_per_s"] = r64["frac"], r64["U"], r64["gathers_per_s"]
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline_txn(kind, rp, done, n_rows, E0, min(E1, E0 + (12 if compact else 60)))
        if ref is not None:
            try:
                r = cpu_reference_tatp(ref, args, dev, C, zipf, shipped=shipped, extra=extra)
            except Exception as ex:  # the reported baseline falls back to the port, with the reason
                r = {"error": f"{type(ex).__name__}: {ex}"}
            finally:
                ref.close()
                if shipped is not None:
                    shipped.close()
            if "error" in r:
                cpu["reference"] = r
            else:
                r["port_on_bench_config"] = cpu
                r["port_on_bench_config_value"] = cpu.get("value")
                cpu = r
        if shipped is not None:
            shipped.close()
        if kind == "tatp" and not args.no_cpu_reference and not compact:
            # the reference's as-shipped per-packet path on this host (its lock_fasst server: the one udp/ server
            # that starts in under a second; tatp's shard server spends minutes populating before it binds) -- and
            # the engine's own socket path (the UDP shim) on the same stream
            from dint_amd import wire as _w
            probe = np.zeros(1 << 18, _w.FASST_MSG)
            probe["type"] = 0
            probe["lid"] = np.random.default_rng(7).integers(0, 24_000_000, len(probe))
            extra["cpu_as_shipped"] = cpu_as_shipped_fasst(probe)
            if extra["cpu_as_shipped"]:
                extra["cpu_as_shipped"]["workload"] = "lock_fasst READ requests over 24M lids (per-packet path cost; not TATP)"
            if not args.no_shim:
                extra["shim_loopback"] = shim_loopback(probe, dev)
    if kind == "tatp":
        dist_name = f"Zipf-{theta}" if zipf else "tatp_nurand (reference)"
        what = (f"TATP full txn mix (35/35/10/2/14/2/2) on {world} MI355X: {n_rows} subscribers ({n_rows // world} per GPU), 3 replicated "
                f"shard servers per GPU group, {C} closed-loop clients per GPU, s_id ~ {dist_name}; "
                f"1 step = {B} epochs = {3 * B} request batches")
        metric = "Mtxn/s + p50/p99 batch latency, TATP"
        rows_key = "subscribers"
    else:
        dist_name = f"Zipf-{theta}" if zipf else "90% of txns on the 4% hot accounts (reference)"
        what = (f"SmallBank (6 txns, 15/15/15/25/15/15, 2PL) on {world} MI355X: {n_rows} accounts hash-sharded, 3 replicated "
                f"shard servers per GPU group, {C} closed-loop clients per GPU, accounts ~ {dist_name}; "
                f"1 step = {B} epochs = {3 * B} request batches")
        metric = "Mtxn/s + p50/p99 batch latency, SmallBank"
        rows_key = "accounts"
    nt = 7 if kind == "tatp" else 6
    commit_rate = stats["committed"] / max(1, stats["txns"])
    out = {
        "metric": metric,
        "value": round(value, 3), "unit": "Mtxn/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "host_issue_ms_per_step": round(t_issue / K * 1e3, 5),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": what, rows_key: n_rows, "clients_per_gpu": C, "epochs_per_step": B,
                   f"{rows_key}_per_gpu": n_rows // world,
                   "scaling_rule": ("--subscribers is PER GPU since r04 (rows = subscribers x N: weak scaling of clients and rows)"
                                    if kind == "tatp" else "--accounts is the TOTAL over all GPUs (default 10M per GPU = configs[4]'s 80M on 8)"),
                   "requests_per_step": round(ops / K / world), "parallelism": f"3 shard servers x hash-shard x{world}",
                   "transport": transport, "exchange_slot_caps": caps,
                   "replay": ("every batch answered IN PLACE (the reply is the request struct mutated, tatp/udp/server_shard.cc:116-121); the "
                              "receive buffers are refilled from pristine copies outside the timed region" if rp.inplace else
                              "separate reply buffers (the partition kernel copies request -> reply)"),
                   "look_ahead": bool(rp.ahead),
                   # scalar copies of the legs the north star is judged on (the driver's record keeps `config`'s scalars)
                   "goodput_Mtxn_s": round(value * commit_rate, 3),
                   "closed_loop_Mtxn_s": closed["value"] if closed else None,
                   "value_pcie_Mtxn_s": host.get("value_pcie"),
                   "latency_p50_us": pct(lat, 50), "latency_p99_us": pct(lat, 99)},
        "Mops_s": round(ops / dt / 1e6, 3), "ops_per_txn": round(ops / max(1.0, txns), 3), "ms_per_epoch": round(dt / (E1 - E0) * 1e3, 5),
        # the reference client's two rates (tatp/caladan/client_udp_shard.cc:102-103): throughput = finished
        # transactions, goodput = committed ones (a NOT_EXIST read of a row the population never made counts as not
        # committed there too)
        "goodput_Mtxn_s": round(value * commit_rate, 3), "abort_rate": round(1.0 - commit_rate, 5),
        "txns_by_type": stats["by_type"][:nt], "committed_by_type": stats["committed_by_type"][:nt],
        "commit_rate_by_type": [round(c / max(1, t), 4) for c, t in zip(stats["committed_by_type"][:nt], stats["by_type"][:nt])],
        "value_repeats": [round(v, 1) for v in repeats], "closed_loop": closed,
        "latency_us": {"p50": pct(lat, 50), "p99": pct(lat, 99), "what": "device side: the 3 batches of one epoch submitted -> replies in HBM"},
        **host, "route_overflow": overflow,
        "roofline": roof, "cpu_baseline": cpu, "setup_s": round(t_setup, 2), **extra,
    }
    del rp, grp
    return out


def client_sweep(args, world, rank, dev, transport, kind):
    """abort rate and Mtxn/s against the number of closed-loop clients (the reference runs 13 machines x <= 300 uthreads,
    exp/run_tatp_wrapper.sh:3-7; 524,288 lock-step clients on 1M Zipf-0.8 subscribers conflict far more often)"""
    import copy
    import gc

    rows = []
    for c in (4096, 32768, 131072, 524288, 1048576, 2097152):  # (2M clients: ~0.97M requests per shard server and epoch -- one pass still)
        a = copy.copy(args)
        a.clients, a.compact, a.steps, a.warmup, a.per_step = c, True, 6, 2, 8
        a.no_cpu_baseline = a.no_rand64 = a.no_closed_loop = a.no_host_path = True
        r = bench_txn(a, world, rank, dev, transport, kind)
        gc.collect()
        if r:
            rows.append({"clients": c, "Mtxn_s": r["value"], "goodput_Mtxn_s": r["goodput_Mtxn_s"], "abort_rate": r["abort_rate"],
                         "commit_rate_by_type": r["commit_rate_by_type"], "ms_per_epoch": r["ms_per_epoch"],
                         "latency_us": r["latency_us"]})
    return rows


def other_workloads(args, world, rank, dev, transport):
    """Compact legs of the other BASELINE configs appended to the default (tatp) line, after its timed region: lock_fasst
    configs[1], lock_2pl, log_server, store configs[2], smallbank (one GPU's 10M-account slice of configs[4]) -- each with
    its own value, roofline and oracle parity on its own bench stream."""
    import copy
    import gc

    out = {}
    for wl in ("tatp_nurand", "tatp_1m_clients", "tatp_2m_clients", "fasst", "fasst_36m", "2pl", "log", "store", "smallbank"):
        a = copy.copy(args)
        a.workload, a.compact, a.steps, a.warmup, a.per_step, a.theta = wl, True, 8, 2, (4 if wl in ("store", "smallbank", "tatp_nurand") else 16), None
        if wl in ("tatp_1m_clients", "tatp_2m_clients"):  # the headline at twice / four times the clients (the r03 sweep was still rising at 524k)
            a.workload, a.clients, a.steps, a.warmup, a.per_step, a.no_cpu_baseline = "tatp", (1 << 20) if wl == "tatp_1m_clients" else (1 << 21), 4, 1, 4, True
        a.no_rand64 = a.no_closed_loop = a.no_host_path = True
        if wl == "fasst_36m":  # the reference's own table size: 288 MB, HBM-resident -- configs[1]'s 1M slots (8 MB) live in L2
            a.workload, a.slots, a.no_cpu_baseline = "fasst", 36_000_000, True
        if wl == "tatp_nurand":  # the headline workload with the reference's OWN key distribution (tatp_nurand, tatp/udp/tatp.h:40-43:
            a.workload, a.theta = "tatp", 0.0  # no hot subscriber), same tables, same clients, oracle parity on its stream
        try:
            r = run_workload(a, world, rank, dev, transport)
        except Exception as ex:  # one leg failing must not take the headline down; it fails the run at the end
            out[wl] = {"error": f"{type(ex).__name__}: {ex}"}
            continue
        gc.collect()
        if r is None:
            continue
        cb = r.get("cpu_baseline") or {}
        par = cb.get("oracle_parity") or (cb.get("port_on_bench_config") or {}).get("oracle_parity")
        out[wl] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
                   "goodput_Mtxn_s": r.get("goodput_Mtxn_s"), "late": r.get("late"),
                   "roofline": {k: r["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "frac")} if r.get("roofline") else None,
                   "kernels_us": r.get("kernels_us"), "latency_us": r.get("latency_us"),
                   "replay_equals_recorded": r.get("replay_equals_recorded"), "abort_rate": r.get("abort_rate"),
                   "cpu_baseline": {k: cb.get(k) for k in ("value", "unit", "cores", "kind")} if cb else None, "oracle_parity": par}
        for k in ("pass_1m", "mixes"):  # log_server: 1M-request passes; store: the 100/0 and 50/50 read/write mixes
            if r.get(k) is not None:
                out[wl][k] = r[k]
    return out


def run_workload(args, world, rank, dev, transport):
    if args.workload in ("fasst", "2pl"):
        return bench_lock(args, world, rank, dev, transport, args.workload)
    if args.workload == "log":
        return bench_log(args, world, rank, dev, transport)
    if args.workload == "store":
        return bench_store(args, world, rank, dev, transport)
    return bench_txn(args, world, rank, dev, transport, args.workload)


def parity_failures(out, path=""):
    """every parity flag of the line that is false: a run whose replies differ from the recorded run, the oracle or the
    reference must not pass as a result"""
    bad = []
    if isinstance(out, dict):
        for k, v in out.items():
            p = f"{path}.{k}" if path else k
            if k in ("oracle_parity", "reference_parity") and isinstance(v, dict) and v.get("ok") is False:
                bad.append(p)
            elif k in ("pcie_parity_ok", "replay_equals_recorded", "equals_host_driver_run", "equals_host_run_of_two_drivers",
                       "replies_equal_64k_batches", "replies_equal") and v is False:
                bad.append(p)
            elif k == "error" and path.startswith("other_workloads"):
                bad.append(p)
            else:
                bad += parity_failures(v, p)
    return bad


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    import gc

    import torch.distributed as dist

    world, rank, dev, transport = init_dist(args)
    out = run_workload(args, world, rank, dev, transport)
    gc.collect()
    if args.workload in ("tatp", "smallbank") and args.sweep_clients and world == 1:
        sweep = client_sweep(args, world, rank, dev, transport, args.workload)
        if out is not None:
            out["client_sweep"] = sweep
    if args.workload == "tatp" and world == 1 and not args.no_other_workloads and not args.force_exchange:
        ow = other_workloads(args, world, rank, dev, transport)
        if out is not None:
            out["other_workloads"] = ow
    if args.workload == "tatp" and world == 1 and not args.no_exchange_leg and not args.force_exchange:
        # the multi-GPU machinery on this one GPU (pack -> "all-to-all" = a device copy -> segments -> back -> unpack), compact:
        # what a rank of an N-GPU run does per epoch besides the engines' passes
        import copy

        a = copy.copy(args)
        a.force_exchange, a.compact, a.steps, a.warmup = True, True, 8, 2
        a.no_rand64 = a.no_closed_loop = a.no_host_path = a.no_cpu_baseline = True
        try:
            r = run_workload(a, world, rank, dev, transport)
            ex = {k: r.get(k) for k in ("value", "unit", "ms_per_step", "ms_per_epoch", "host_issue_ms_per_step", "route_overflow",
                                        "latency_us", "replay_equals_recorded")}
            ex["what"] = "--force-exchange on one GPU: the routing kernels and both exchange legs (device copies) around the same passes"
            if out is not None and out.get("value"):
                ex["ratio_to_value"] = round(r["value"] / out["value"], 3)
                out["config"]["exchange_Mtxn_s"], out["config"]["exchange_ratio_to_value"] = r["value"], ex["ratio_to_value"]
        except Exception as e_:
            ex = {"error": f"{type(e_).__name__}: {e_}"}
        gc.collect()
        if out is not None:
            out["exchange"] = ex
    bad = []
    if rank == 0:
        bad = parity_failures(out)
        if bad:
            out["parity_failures"] = bad
            out["value_unchecked"], out["value"] = out["value"], None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        sys.exit(3)


if __name__ == "__main__":
    main()
