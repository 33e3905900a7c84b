"""File-per-GPU sharding helpers: captures are independent, so rank r simply owns
captures [r*S, (r+1)*S) and there is no data-path collective.  torch.distributed is used only for
the barrier and for max-over-ranks of the elapsed time (backend "nccl" = RCCL on GPUs, "gloo" in
the CPU tests)."""
import os

BASE_SEED = 0xC0FFEE


def rank_env():
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0)))


def capture_seed(rank, streams_per_rank, s):
    """Seed of capture `s` of rank `rank`: globally unique, rank 0 reproduces the 1-GPU workload."""
    return BASE_SEED + rank * streams_per_rank + s


def device_of(file_index, n_devices):
    """File-per-GPU map of the batch CLI (`rtl_wmbus_hip -G all|list FILE...`, wm_main.c shard_slot) and of
    SURVEY.md 8(e): stream s -> position s mod n in the device list."""
    return file_index % n_devices


def owned_captures(rank, world, total):
    """Contiguous block partition of `total` captures (used when a fixed file list is sharded)."""
    per = (total + world - 1) // world
    return range(min(total, rank * per), min(total, (rank + 1) * per))


def init(world, local_rank, backend=None):
    """Process group for barrier / max-reduce only; returns the torch.distributed module or None."""
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def barrier(dist):
    if dist is None:
        return
    import torch
    dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def _reduce(dist, value, op):
    import torch
    dev = "cuda" if (dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([float(value)], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(dist, value):
    return value if dist is None else _reduce(dist, value, dist.ReduceOp.MAX)


def sum_over_ranks(dist, value):
    return value if dist is None else _reduce(dist, value, dist.ReduceOp.SUM)
