"""Multi-GPU sharding of the frame path: frames are independent units, so the path
shards by camera stream / by contiguous frame ranges with NO collective on the data path.  The only
communication is the trivial frame scatter when a batch originates on one rank, the broadcast of
per-camera constants, the barrier and the max-over-ranks time of the benchmark -- all through
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests)."""
import os

import torch
import torch.distributed as dist


def _solo():
    """True when there is nobody to talk to: no process group, or a group of one -- unless RIP_DIST_FORCE=1 asks for the
    collectives to run anyway (a single-rank RCCL communicator on a 1-GPU box: the only way the nccl code paths below --
    device tensors, dtypes, the calls themselves -- execute before an 8-GPU node does; tests/test_configs_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and os.environ.get("RIP_DIST_FORCE", "") != "1"


def streams_of_rank(n_streams, world_size, rank):
    """Camera c is owned by rank c mod world_size: its Kalman state, maps and LUTs stay resident there."""
    return list(range(rank, n_streams, world_size))


def frame_range_of_rank(n_frames, world_size, rank):
    """Contiguous, balanced [start, stop) range of a single-stream batch (temporal consistency off)."""
    base, extra = divmod(n_frames, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _dev(t):
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return t.cuda()
    return t


def max_over_ranks(value):
    """The benchmark's time is the slowest rank's."""
    if _solo():
        return float(value)
    t = _dev(torch.tensor([float(value)], dtype=torch.float64))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value):
    if _solo():
        return float(value)
    t = _dev(torch.tensor([float(value)], dtype=torch.float64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value):
    """Every rank's value, in rank order (all_gather): the per-rank record of the benchmark line."""
    if _solo():
        return [float(value)]
    mine = _dev(torch.tensor([float(value)], dtype=torch.float64))
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def communicator_census():
    """What the process group itself reports, plus what a collective over it proves: {"backend", "world_size" (as the
    communicator reports it), "ranks_counted" (all_reduce SUM of one per rank), "devices" (CUDA ordinal of every rank, -1 on
    a host backend)} -- so that a multi-GPU benchmark record answers 'did RCCL see N ranks on N devices' by itself."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "ranks_counted": 1, "devices": [torch.cuda.current_device() if torch.cuda.is_available() else -1]}
    one = _dev(torch.ones(1, dtype=torch.int64))
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_counted": int(one.item()),
            "devices": [int(v) for v in gather_over_ranks(dev)]}


def broadcast_constants(tensor, src=0):
    """Per-camera constants (undistortion maps, LUTs, ccc model spectrum) computed once on `src`."""
    if not _solo():
        dist.broadcast(tensor, src=src)
    return tensor


def scatter_frames(batch, frame_shape, dtype=torch.uint8, src=0, device=None):
    """Distributes a batch [n, ...] resident on `src` so that every rank receives its
    frame_range_of_rank() slice.  Ranks other than `src` pass batch=None.  Returns this rank's frames.

    This is the 'trivial frame scatter' of the north star: it is bounded by the source GPU's xGMI
    egress (7 links x ~153 GB/s), far below the processing rate, so steady-state benchmarks keep the
    frames resident per GPU and report the scatter separately."""
    if _solo():
        return batch
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([batch.shape[0] if rank == src else 0], dtype=torch.int64)
    n = _dev(n)
    dist.broadcast(n, src=src)
    n = int(n.item())
    start, stop = frame_range_of_rank(n, world, rank)
    dev = device if device is not None else (batch.device if batch is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    mine = torch.empty((stop - start,) + tuple(frame_shape), dtype=dtype, device=dev)
    # equal-size scatter needs padding; ranges differ by at most one frame, so use point-to-point
    if rank == src:
        reqs = []
        for r in range(world):
            a, b = frame_range_of_rank(n, world, r)
            if r == src:
                mine.copy_(batch[a:b])
            elif b > a:
                reqs.append(dist.isend(batch[a:b].contiguous(), dst=r))
        for q in reqs:
            q.wait()
    elif stop > start:
        dist.recv(mine, src=src)
    return mine
