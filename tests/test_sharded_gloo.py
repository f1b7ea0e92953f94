"""world_size-2 CPU test of the multi-GPU exchange (gloo): routing + inverse routing around a
per-rank server double must reproduce the single-server serial replay of the rank-major
concatenation of the ingest slices."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nslots, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tracegen
    from dint_amd import wire
    from dint_amd.sharded import ShardedEngine
    from oracle import oracle as orc

    msg = wire.FASST_MSG.itemsize
    n = 5000
    # the server double of this rank: an oracle over the LOCAL slots, fed with lids remapped so
    # that local_slot = global_slot // world  (what the sharded engine does on the GPU)
    full = orc.FasstOracle(nslots)  # per-rank replica used only through slots that are "home" here

    def home_fn(req2d):
        lids = np.frombuffer(req2d.numpy().tobytes(), wire.FASST_MSG)["lid"]
        h = np.array([orc.fasthash64(int(l).to_bytes(4, "little")) % nslots % world for l in lids], np.uint8)
        return torch.from_numpy(h)

    def local_fn(recv2d):
        m = np.frombuffer(recv2d.numpy().tobytes(), wire.FASST_MSG)
        out = full.replay(m)
        recv2d.copy_(torch.from_numpy(np.frombuffer(out.tobytes(), np.uint8).reshape(-1, msg).copy()))

    sh = ShardedEngine(None, world, rank, msg_size=msg, home_fn=home_fn, local_fn=local_fn)
    # a second, independent server double replays the same steps with the split sizes the first run discovered
    # (the host-sync-free form bench.py uses for a recorded trace)
    full2 = orc.FasstOracle(nslots)

    def local_fn2(recv2d):
        m = np.frombuffer(recv2d.numpy().tobytes(), wire.FASST_MSG)
        recv2d.copy_(torch.from_numpy(np.frombuffer(full2.replay(m).tobytes(), np.uint8).reshape(-1, msg).copy()))

    sh2 = ShardedEngine(None, world, rank, msg_size=msg, home_fn=home_fn, local_fn=local_fn2)
    outs = []
    for step in range(3):
        req = tracegen.fasst_random(n, seed=100 * step + rank, n_hot=16, p_hot=0.8)
        d_req = torch.from_numpy(np.frombuffer(req.tobytes(), np.uint8).copy())
        d_rep = torch.empty_like(d_req)
        splits = sh.submit_device(d_req, n, d_rep)
        outs.append(d_rep.numpy().tobytes())
        d_rep2 = torch.empty_like(d_req)
        assert sh2.submit_device(d_req, n, d_rep2, splits=splits) == splits
        assert d_rep2.numpy().tobytes() == outs[-1]
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_exchange_world2_gloo():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tracegen
    from dint_amd import wire
    from oracle import oracle as orc

    world, nslots = 2, 4801
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nslots, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-server serial replay of the rank-major concatenation, step by step
    o = orc.FasstOracle(nslots)
    for step in range(3):
        reqs = [tracegen.fasst_random(5000, seed=100 * step + r, n_hot=16, p_hot=0.8) for r in range(world)]
        want = o.replay(np.concatenate(reqs))
        for r in range(world):
            assert res[r][step] == want[r * 5000:(r + 1) * 5000].tobytes(), (step, r)
