#!/usr/bin/env python3
"""bench.py -- the DINT server hot path on MI355X, one step = one batch of synthetic wire messages
already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tatp|fasst] ...

Default workload (BASELINE.json metric "Mtxn/s ... TATP Zipf-0.8", configs[3]): the TATP transaction
mix against 1M subscribers.  W virtual clients run the reference client's seven transactions in lock
step (dint_amd/csrc/txn_driver.cc); one step = one EPOCH = every client's current phase, i.e. three
request batches (one per replicated shard server, as in the reference's 3-server deployment), which the
three engines of the GPU process.  The closed loop is first run once and recorded; the timed region
replays the recorded batches from HBM (no host work, no PCIe) and the replies are checked byte for byte
against the recorded run.  `--workload fasst` = BASELINE configs[1] (lock_fasst, 1M slots, 64k batches).

Rank 0 prints ONE JSON line; DESIGN.md "Measurement" defines every field.  For N > 1 this file is
launched by torch.distributed.run, one rank per GPU: every logical shard server is hash-partitioned
over the ranks and requests are routed to their home GPU with an all-to-all over RCCL.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 65536
KV_PASS = 1 << 20  # requests per kernel pass of the store / tatp / smallbank engines (dint_config.max_pass = 0)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("DINT_BENCH_WORKLOAD", "tatp"), choices=["tatp", "fasst"])
    ap.add_argument("--slots", type=int, default=1 << 20, help="lock_fasst table slots (BASELINE configs[1]: 1M)")
    ap.add_argument("--theta", type=float, default=0.8, help="Zipf skew of the key stream; 0 = the reference's own distribution")
    ap.add_argument("--subscribers", type=int, default=1_000_000, help="tatp subscribers (BASELINE configs[3]: 1M)")
    ap.add_argument("--clients", type=int, default=524288,
                    help="tatp closed-loop clients per GPU (one outstanding request each: a shard server sees ~clients * 0.46 "
                         "requests per epoch, all resolved in one kernel pass)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rand64", action="store_true")
    return ap.parse_args()


def init_dist():
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    return world, rank, torch.cuda.current_device()


def barrier(world):
    import torch
    import torch.distributed as dist

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(dt, world):
    import torch
    import torch.distributed as dist

    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def sum_over_ranks(x, world):
    import torch
    import torch.distributed as dist

    if world > 1:
        t = torch.tensor([float(x)], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        x = float(t.item())
    return x


def pmc_traffic(kernels):
    """HBM bytes per launch of `kernels` from the committed rocprofv3 --pmc summary of this same command
    (profiles/r01_tatp_rocprofv3_summary.txt: FETCH_SIZE + WRITE_SIZE, KiB per dispatch, separate passes as
    MI355X_MICROARCH.md prescribes; the rows over the last dispatches = the replayed epochs, without the population
    passes).  Raw counter values: the guide's x2 FETCH_SIZE correction applies to wide coalesced streams, the
    resolve kernel issues 8..64-byte random accesses, for which it is uncalibrated."""
    path = os.path.join(ROOT, "profiles", "r01_tatp_rocprofv3_summary.txt")
    try:
        tot = {False: 0.0, True: 0.0}  # [rows over all dispatches, rows over the last dispatches]
        last = False
        for line in open(path):
            if line.startswith("kernel"):
                last = "(last" in line
            f = line.split()
            if len(f) >= 4 and f[1] in ("FETCH_SIZE", "WRITE_SIZE") and f[0].split("<")[0] in kernels:
                tot[last] += float(f[3]) * 1024.0
        t = tot[True] or tot[False]
        return int(t) if t else None
    except OSError:
        return None


def rocprof_avg_us(kernels):
    """Average begin->end duration of `kernels` over the same launches in the committed rocprofv3 --kernel-trace summary
    (first block of profiles/r01_tatp_rocprofv3_summary.txt, rows over the last dispatches).  The live HIP-event interval
    (`kernel_avg_us`) additionally contains the dispatch gap on a stream that shares the GPU with two other engines."""
    path = os.path.join(ROOT, "profiles", "r01_tatp_rocprofv3_summary.txt")
    try:
        tot, last, seen = 0.0, False, set()
        for line in open(path):
            if line.startswith("#") and seen:
                break  # only the kernel-trace block
            if line.startswith("kernel"):
                last = "(last" in line
            f = line.split()
            if last and len(f) == 5 and f[0].split("<")[0] in kernels and f[0] not in seen:
                seen.add(f[0])
                tot += float(f[2])
        return round(tot, 3) if seen else None
    except (OSError, ValueError):
        return None


def rand64(extra, value_ops_per_s, dev):
    """Measured random-64B HBM roofline: gathers over an 8 GiB table."""
    from dint_amd.engine import bench_rand64

    try:
        aps, _ = bench_rand64(8 << 30, 1 << 28, False, dev)
        aps_w, _ = bench_rand64(8 << 30, 1 << 28, True, dev)
        extra["rand64_Gaccess_s"] = round(aps / 1e9, 3)
        extra["rand64_rw_Gaccess_s"] = round(aps_w / 1e9, 3)
        extra["ops_frac_of_rand64"] = round(value_ops_per_s / aps, 5)
    except Exception as ex:  # measurement helper only
        extra["rand64_error"] = str(ex)


# ------------------------------------------------------------------------------------------- lock_fasst
def cpu_baseline_fasst(sample: np.ndarray, nslots: int):
    """The CPU baseline on rank 0's host cores over a bounded sample of the same stream: the unmodified
    reference server when its replay binary is present (kind "reference"), else the C restatement."""
    from oracle import oracle as orc

    if nslots == 36_000_000 and orc.ref_available("lock_fasst"):
        _, st = orc.ref_replay("lock_fasst", sample)
        return {"value": st["ops_per_s"] / 1e6, "unit": "Mtxn/s", "cores": 1, "kind": "reference",
                "sample": f"{len(sample)} requests of the bench stream, unmodified lock_fasst/udp/server.cc, sockets interposed"}
    o = orc.FasstOracle(nslots)
    t = time.perf_counter()
    o.replay(sample)
    dt = time.perf_counter() - t
    return {"value": len(sample) / dt / 1e6, "unit": "Mtxn/s", "cores": 1, "kind": "port",
            "sample": f"{len(sample)} requests of the bench stream, oracle/dint_oracle.c ({nslots} slots)"}


def bench_fasst(args, world, rank, dev):
    import torch

    from dint_amd import wire, workloads
    from dint_amd.engine import Engine
    from dint_amd.sharded import ShardedEngine

    K, W = args.steps, args.warmup
    # FaSST-client-shaped stream, Zipf(theta) keys over 24M lids, 4096 interleaved virtual clients;
    # every rank ingests its own slice
    n_req = BATCH * (K + W)
    stream = workloads.fasst_stream(n_req, key_space=24_000_000, theta=args.theta, seed=1234 + rank)
    stream = workloads.interleave(stream, 4096)
    n_req = len(stream) // BATCH * BATCH
    assert n_req // BATCH >= K + W
    d_req = torch.from_numpy(np.frombuffer(stream[:n_req].tobytes(), np.uint8).copy()).cuda()
    d_rep = torch.empty_like(d_req)
    msg = wire.FASST_MSG.itemsize

    eng = Engine(wire.Workload.FASST, n_slots=args.slots, device=dev, shard_index=rank, shard_count=world)
    sh = ShardedEngine(eng, world, rank) if world > 1 else None
    st = torch.cuda.current_stream().cuda_stream

    def step(b):
        lo = b * BATCH * msg
        if sh is None:
            eng.submit_device(d_req.data_ptr() + lo, BATCH, d_rep.data_ptr() + lo, st)
        else:
            sh.submit_device(d_req[lo:lo + BATCH * msg], BATCH, d_rep[lo:lo + BATCH * msg])

    for b in range(W):
        step(b)
    barrier(world)
    t0 = time.perf_counter()
    for b in range(W, W + K):
        step(b)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)

    lat = []
    for b in range(W, W + min(K, 100)):
        torch.cuda.synchronize()
        t = time.perf_counter()
        step(b)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.array(lat)

    roof, extra = None, {}
    if sh is None:
        eng.timing_enable(True)
        for b in range(W, W + min(K, 200)):
            step(b)
        torch.cuda.synchronize()
        tim = eng.timing_read()
        eng.timing_enable(False)
        extra["kernels_us"] = {k: round(v["avg_us"], 3) for k, v in tim.items()}
        mut = float((stream[W * BATCH:(W + K) * BATCH]["type"] != 0).mean())
        # 9 (req) + 9 (reply) + 8 (lock+ver read) + 8 if the op mutates the slot
        alg_bytes = BATCH * (26.0 + 8.0 * mut)
        dom = max(tim.items(), key=lambda kv: kv[1]["avg_us"])
        achieved = alg_bytes / (dom[1]["avg_us"] * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "alg_bytes_per_launch": int(alg_bytes), "kernel_avg_us": round(dom[1]["avg_us"], 3)}

    value = world * K * BATCH / dt / 1e6
    if rank != 0:
        return None
    if not args.no_rand64 and sh is None:
        rand64(extra, value * 1e6, dev)
    cpu = None if args.no_cpu_baseline else cpu_baseline_fasst(stream[W * BATCH:W * BATCH + 4_000_000].copy(), args.slots)
    return {
        "metric": "Mtxn/s (lock_fasst: 1 txn = 1 request) + p50/p99 batch latency",
        "value": round(value, 3), "unit": "Mtxn/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"lock_fasst on {world} MI355X: {args.slots}-slot lock table, 64k-request batches, "
                               f"Zipf-{args.theta} over 24M lids, FaSST client op mix (read proportion 0.8)",
                   "batch": BATCH, "slots": args.slots, "parallelism": f"hash-shard x{world}"},
        "latency_us": {"p50": round(float(np.percentile(lat, 50)), 2), "p99": round(float(np.percentile(lat, 99)), 2)},
        "roofline": roof, "cpu_baseline": cpu, **extra,
    }


# ------------------------------------------------------------------------------------------------ tatp
# algorithmic bytes per request: 55 (request) + 55 (reply) + the row / lock bytes the op needs
TATP_ALG = {0: 162, 1: 126, 2: 126, 12: 170, 18: 170, 22: 170, 13: 162, 19: 162, 23: 162, 14: 174, 24: 174}
TATP_LOG_TYPES = (14, 24)


def cpu_baseline_tatp(trace, done, n_sub, lo, hi):
    """Shard server 0's recorded request stream of epochs [lo, hi) replayed on one host core by the CPU
    port (oracle/dint_oracle.c).  A transaction costs `ops_per_txn` requests over the three servers, so one
    core serving all three streams back to back completes (ops/s) / ops_per_txn transactions per second."""
    from oracle import oracle as orc

    o = orc.TatpOracle(n_sub)
    for e in range(lo):  # bring the replica to the state at the start of the sample (not timed)
        o.replay(trace[e][0][0])
    n = 0
    t = time.perf_counter()
    for e in range(lo, hi):
        o.replay(trace[e][0][0])
        n += len(trace[e][0][0])
    dt = time.perf_counter() - t
    ops_all = sum(sum(len(r) for r in trace[e][0]) for e in range(lo, hi))
    ops_per_txn = ops_all / max(1, sum(done[lo:hi]))
    return {"value": round(n / dt / ops_per_txn / 1e6, 4), "unit": "Mtxn/s", "cores": 1, "kind": "port",
            "ops_per_s": round(n / dt), "ops_per_txn": round(ops_per_txn, 3),
            "sample": f"{n} requests = shard server 0's stream of {hi - lo} bench epochs, oracle/dint_oracle.c "
                      f"({n_sub} subscribers), 1 thread"}


def bench_tatp(args, world, rank, dev):
    import torch

    from dint_amd import wire
    from dint_amd.driver import Driver
    from dint_amd.replay import Replay, ShardGroup, record

    K, W = args.steps, args.warmup
    n_sub, C = args.subscribers, args.clients
    theta = args.theta if args.theta > 0 else None
    t_setup = time.perf_counter()
    grp = ShardGroup(wire.Workload.TATP, n_sub, device=dev, rank=rank, world=world)
    grp.sync()
    grp.snapshot()
    drv = Driver(wire.Workload.TATP, C, n_sub, first_client=rank * C, zipf_theta=theta)
    trace, done = record(drv, grp, W + K)  # the closed loop, once, through the real engines
    stats = drv.stats()
    grp.sync()
    grp.restore()
    rp = Replay(trace, grp.msg)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    rp.run(grp, 0, W)
    grp.sync()
    barrier(world)
    t0 = time.perf_counter()
    rp.run(grp, W, W + K)
    grp.sync()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    rp.check(0, W + K)  # parity with the recorded closed-loop run, every reply byte

    txns = sum_over_ranks(sum(done[W:W + K]), world)
    ops = sum_over_ranks(rp.ops(W, W + K), world)
    value = txns / dt / 1e6

    # per-epoch latency: submit of the three batches -> all replies visible in HBM
    grp.restore()
    lat = []
    for e in range(min(W + K, 120)):
        grp.sync()
        t = time.perf_counter()
        rp.run(grp, e, e + 1)
        grp.sync()
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.array(lat[min(W, len(lat) // 2):])

    roof, extra = None, {}
    if world == 1:
        grp.restore()
        for e in grp.engines:
            e.timing_enable(True)
        n_t = min(W + K, 200)
        rp.run(grp, 0, n_t)
        grp.sync()
        tims = [e.timing_read() for e in grp.engines]
        for e in grp.engines:
            e.timing_enable(False)
        names = list(tims[0].keys())
        avg = {k: float(np.mean([t[k]["avg_us"] for t in tims])) for k in names}
        extra["kernels_us"] = {k: round(v, 3) for k, v in avg.items()}
        # The table requests of a pass are resolved by k_kv_resolve (every bin of the pass, one launch): algorithmic
        # bytes of the table requests over its duration.  Log requests are finished by k_kv_place.
        resolve_us = avg.get("k_kv_resolve", 0.0)
        scatter_us = avg.get("k_kv_count", 0.0) + avg.get("k_kv_scan", 0.0) + avg.get("k_kv_place", 0.0)
        dom = "k_kv_resolve" if resolve_us >= scatter_us else "k_kv_count+k_kv_scan+k_kv_place"
        dom_us = max(resolve_us, scatter_us)
        tot_b, launches = 0.0, 0
        for e in range(n_t):
            for s in range(3):
                ty = trace[e][0][s]["type"]
                if len(ty) == 0:
                    continue
                launches += -(-len(ty) // KV_PASS)
                for code, b in TATP_ALG.items():
                    if resolve_us < scatter_us or code not in TATP_LOG_TYPES:
                        tot_b += b * int((ty == code).sum())
        alg = tot_b / max(1, launches)
        achieved = alg / (dom_us * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(dom.split("+")),
                "alg_bytes_per_launch": int(alg), "kernel_avg_us": round(dom_us, 3),
                "kernel_avg_us_rocprofv3": rocprof_avg_us(dom.split("+"))}
    if rank != 0:
        return None
    if not args.no_rand64 and world == 1:
        rand64(extra, ops / dt, dev)
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline_tatp(trace, done, n_sub, W, W + min(K, 60))
    dist_name = f"Zipf-{args.theta}" if theta else "tatp_nurand (reference)"
    return {
        "metric": "Mtxn/s + p50/p99 batch latency, TATP",
        "value": round(value, 3), "unit": "Mtxn/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"TATP full txn mix (35/35/10/2/14/2/2) on {world} MI355X: {n_sub} subscribers, 3 replicated "
                               f"shard servers per GPU group, {C} closed-loop clients per GPU, s_id ~ {dist_name}; "
                               f"1 step = 1 epoch = 3 request batches",
                   "subscribers": n_sub, "clients_per_gpu": C, "requests_per_step": round(ops / K / world),
                   "parallelism": f"3 shard servers x hash-shard x{world}"},
        "Mops_s": round(ops / dt / 1e6, 3), "ops_per_txn": round(ops / max(1.0, txns), 3),
        "abort_rate": round(1.0 - stats["committed"] / max(1, stats["txns"]), 5),
        "txns_by_type": stats["by_type"][:7], "committed_by_type": stats["committed_by_type"][:7],
        "latency_us": {"p50": round(float(np.percentile(lat, 50)), 2), "p99": round(float(np.percentile(lat, 99)), 2)},
        "roofline": roof, "cpu_baseline": cpu, "setup_s": round(t_setup, 2), **extra,
    }


def main():
    args = parse()
    import torch.distributed as dist

    world, rank, dev = init_dist()
    out = bench_tatp(args, world, rank, dev) if args.workload == "tatp" else bench_fasst(args, world, rank, dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
