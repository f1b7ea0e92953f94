#!/usr/bin/env python3
"""bench.py -- one pass of the DINT hot path per step over HBM-resident batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fasst|...]

A step = one 65,536-request batch of synthetic wire messages, already resident in HBM,
through the engine (dint_submit_device).  Rank 0 prints ONE JSON line (see DESIGN.md
"Measurement" for how every field is defined).  For N > 1 this file is launched by
torch.distributed.run, one rank per GPU; each rank ingests its own batch slice, requests
are routed to their home shard with an all-to-all over RCCL and the replies routed back.
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=os.environ.get("DINT_BENCH_WORKLOAD", "fasst"))
    ap.add_argument("--slots", type=int, default=1 << 20, help="lock_fasst table slots (BASELINE configs[1]: 1M)")
    ap.add_argument("--theta", type=float, default=0.8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rand64", action="store_true")
    return ap.parse_args()


def cpu_baseline_fasst(sample: np.ndarray, nslots: int):
    """Time the CPU baseline on rank 0's host cores over a bounded sample of the same stream:
    the unmodified reference server when its replay binary is present (kind "reference"),
    else the C restatement (kind "port").  One thread, as the serial oracle."""
    from oracle import oracle as orc

    if nslots == 36_000_000 and orc.ref_available("lock_fasst"):
        _, st = orc.ref_replay("lock_fasst", sample)
        return {"value": st["ops_per_s"] / 1e6, "unit": "Mops/s", "cores": 1, "kind": "reference",
                "sample": f"{len(sample)} requests of the bench stream, unmodified lock_fasst/udp/server.cc, sockets interposed"}
    o = orc.FasstOracle(nslots)
    t = time.perf_counter()
    o.replay(sample)
    dt = time.perf_counter() - t
    return {"value": len(sample) / dt / 1e6, "unit": "Mops/s", "cores": 1, "kind": "port",
            "sample": f"{len(sample)} requests of the bench stream, oracle/dint_oracle.c ({nslots} slots)"}


def main():
    args = parse()
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
    dev = torch.cuda.current_device()

    from dint_amd import wire, workloads
    from dint_amd.engine import Engine, bench_rand64
    from dint_amd.sharded import ShardedEngine

    K, W = args.steps, args.warmup
    if args.workload != "fasst":
        raise SystemExit(f"workload {args.workload} not wired into bench.py yet")

    # ---- synthetic input: FaSST-client-shaped stream, Zipf(theta) keys over 24M lids,
    # 4096 interleaved virtual clients; every rank ingests its own slice ----
    n_req = BATCH * (K + W)
    stream = workloads.fasst_stream(n_req, key_space=24_000_000, theta=args.theta, seed=1234 + rank)
    stream = workloads.interleave(stream, 4096)
    n_req = len(stream) // BATCH * BATCH
    n_batches = n_req // BATCH
    assert n_batches >= K + W
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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for b in range(W):
        step(b)
    barrier()
    t0 = time.perf_counter()
    for b in range(W, W + K):
        step(b)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-batch latency (submit -> replies visible in HBM), one batch at a time
    lat = []
    for b in range(W, W + min(K, 100)):
        torch.cuda.synchronize()
        t = time.perf_counter()
        step(b)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t) * 1e6)
    lat = np.array(lat)

    # ---- dominant-kernel duration with HIP events on the launch stream (same K steps) ----
    roof = None
    extra = {}
    if sh is None:
        eng.timing_enable(True)
        for b in range(W, W + min(K, 200)):
            step(b)
        torch.cuda.synchronize()
        tim = eng.timing_read()
        eng.timing_enable(False)
        extra["kernels_us"] = {k: round(v["avg_us"], 3) for k, v in tim.items()}
        types = stream[W * BATCH:(W + K) * BATCH]["type"]
        mut = float((types != 0).mean())
        # lock_fasst algorithmic bytes per request: 9 (req) + 9 (reply) + 8 (lock+ver
        # read) + 8 if the op mutates the slot
        alg_bytes = BATCH * (26.0 + 8.0 * mut)
        dom = max(tim.items(), key=lambda kv: kv[1]["avg_us"])
        achieved = alg_bytes / (dom[1]["avg_us"] * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": dom[0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "alg_bytes_per_launch": int(alg_bytes), "kernel_avg_us": round(dom[1]["avg_us"], 3)}

    value = world * K * BATCH / dt / 1e6  # whole-job Mops/s (one request = one "txn" on the micro paths)
    if rank == 0:
        if not args.no_rand64 and sh is None:
            try:
                aps, _ = bench_rand64(8 << 30, 1 << 28, False, dev)
                aps_w, _ = bench_rand64(8 << 30, 1 << 28, True, dev)
                extra["rand64_Gaccess_s"] = round(aps / 1e9, 3)
                extra["rand64_rw_Gaccess_s"] = round(aps_w / 1e9, 3)
                extra["frac_of_rand64"] = round(value * 1e6 / aps, 5)
            except Exception as ex:  # measurement helper only
                extra["rand64_error"] = str(ex)
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline_fasst(stream[W * BATCH:W * BATCH + 4_000_000].copy(), args.slots)
        out = {
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
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
