"""The N > 1 path of bench.py on CPU: two gloo processes exercise the sharding helpers (stream / frame
partition, frame scatter, constant broadcast, max-over-ranks time).  No data-path collective exists."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raw_image_pipeline_amd import sharding


def test_partitions_cover_everything_once():
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 7, 8, 64, 513):
            owned = [sharding.streams_of_rank(n, world, r) for r in range(world)]
            flat = sorted(s for o in owned for s in o)
            assert flat == list(range(n))
            ranges = [sharding.frame_range_of_rank(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    # 8-camera rig on 8 GPUs: one stream per GPU (BASELINE config 4)
    assert [sharding.streams_of_rank(8, 8, r) for r in range(8)] == [[r] for r in range(8)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_frames, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # frame scatter from rank 0
        batch = None
        if rank == 0:
            batch = torch.arange(n_frames * 6 * 4, dtype=torch.int64).reshape(n_frames, 6, 4).to(torch.uint8)
        mine = sharding.scatter_frames(batch, (6, 4), dtype=torch.uint8, src=0, device="cpu")
        a, b = sharding.frame_range_of_rank(n_frames, world, rank)
        expect = torch.arange(n_frames * 6 * 4, dtype=torch.int64).reshape(n_frames, 6, 4).to(torch.uint8)[a:b]
        ok_scatter = bool(torch.equal(mine, expect))
        # constants broadcast (e.g. the float2 undistortion maps)
        maps = torch.full((5, 3), float(rank + 1))
        sharding.broadcast_constants(maps, src=0)
        ok_bcast = bool((maps == 1.0).all())
        # benchmark reductions: every rank processes its own frames; the time is the slowest rank's
        t = sharding.max_over_ranks(1.0 + rank)
        total = sharding.sum_over_ranks(float(b - a))
        dist.barrier()
        results[rank] = (ok_scatter, ok_bcast, t, total)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 8])
def test_two_rank_gloo_sharding(n_frames):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_frames, results), nprocs=world, join=True)
    assert len(results) == world
    for rank in range(world):
        ok_scatter, ok_bcast, t, total = results[rank]
        assert ok_scatter and ok_bcast
        assert t == 2.0            # max over ranks of (1 + rank)
        assert total == n_frames   # all frames accounted for exactly once
