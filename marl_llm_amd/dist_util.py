"""Multi-GPU plumbing for the env step: environments are independent, so the step path has NO collective.
One process per GPU owns a contiguous slice of the global env range; torch.distributed (RCCL on GPUs, gloo on
CPU) is used only for barriers, the max-over-ranks of a timing and the optional single host-side gather of
obs / reward the north-star allows."""
import os

import torch
import torch.distributed as dist


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, local_rank=None):
    """Initialise the default process group from the torchrun-style environment (RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT; bench.py sets the same variables when it spawns the ranks itself).  No-op for world size 1.
    `local_rank` overrides LOCAL_RANK as the device ordinal handed to RCCL (one-GPU rehearsals use 0 everywhere)."""
    rank, env_local, world = rank_world()
    local_rank = env_local if local_rank is None else local_rank
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, **kw)
    return rank, local_rank, world


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_range(n_total, rank, world):
    """Contiguous slice [begin, end) of n_total environments owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(t):
    """The one host-side gather of the step outputs: every rank contributes its [E_local, ...] CPU tensor (equal
    E_local on every rank), rank 0 receives the concatenation in rank (= env) order, other ranks get None."""
    if not dist.is_initialized():
        return t
    t = t.contiguous()
    world = dist.get_world_size()
    if dist.get_rank() == 0:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, gather_list=parts, dst=0)
        return torch.cat(parts, dim=0)
    dist.gather(t, gather_list=None, dst=0)
    return None
