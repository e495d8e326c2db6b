"""world_size-2 CPU (gloo) coverage of the N>1 path: stream->rank sharding, the optional descriptor-block
all-gather (config C5) and the max-over-ranks timing used by bench.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from orb_slam3_fast_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_streams, cap = 6, 5
        mine = sharding.assign_streams(n_streams, world, rank)
        # fake per-image descriptor blocks: image of stream s has s+1 valid rows filled with value 10*s + row
        counts = torch.tensor([s + 1 for s in mine], dtype=torch.int32)
        desc = torch.zeros((len(mine), cap, 32), dtype=torch.uint8)
        for i, s in enumerate(mine):
            for r in range(min(s + 1, cap)):
                desc[i, r] = 10 * s + r
        gc, gd = sharding.allgather_descriptor_blocks(counts, desc, cap)
        mc, md = sharding.allgather_members(counts, desc)          # the two collectives the C ABI issues, on the same buffers
        assert torch.equal(mc, gc) and torch.equal(md, gd)
        t = sharding.max_over_ranks(1.0 + rank)
        rate = sharding.whole_job_rate(units_per_rank=4, steps=10, seconds=t)
        q.put((rank, mine, gc.tolist(), gd[:, :, 0].tolist(), t, rate))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_allgather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    streams = [r[1] for r in res]
    assert streams == [[0, 2, 4], [1, 3, 5]]                       # disjoint, complete, round robin
    for rank, mine, gc, gd0, t, rate in res:
        order = streams[0] + streams[1]                             # rank-major order of the gathered blocks
        assert gc == [s + 1 for s in order]
        for blk, s in zip(gd0, order):
            assert blk == [10 * s + r if r < s + 1 else 0 for r in range(5)]
        assert t == 2.0 and rate == 2 * 4 * 10 / 2.0                # max over ranks, whole-job aggregate


def test_pack_roundtrip():
    rng = np.random.default_rng(0)
    counts = torch.tensor(rng.integers(0, 70000, 4), dtype=torch.int32)
    desc = torch.tensor(rng.integers(0, 256, (4, 7, 32)), dtype=torch.uint8)
    c2, d2 = sharding.unpack_descriptor_blocks(sharding.pack_descriptor_blocks(counts, desc, 7), 7)
    assert torch.equal(c2, counts) and torch.equal(d2, desc)
