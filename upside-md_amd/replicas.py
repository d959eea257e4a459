"""Multi-GPU layer: replicas only.

An Upside run has no data-path exchange between independent simulations (the reference runs them as OpenMP
threads over `systems`, /root/reference/src/main.cpp:470-500, 640-700; only replica exchange couples them and it
moves two scalars per pair).  One process drives one GPU; rank r of W owns a contiguous block of the global
system list, seeds its thermostat streams from the GLOBAL system index (main.cpp:459: seed + system index), and
nothing crosses xGMI inside the timed loop.  The only collectives are the barrier around the timed region and
the MAX over ranks of the elapsed time that the bench contract asks for.

Works with any torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
"""
import os


def world_from_env():
    """(rank, local_rank, world) as torch.distributed.run exports them; (0, 0, 1) when launched plainly."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def shard(n_total, world, rank):
    """[start, stop) of the global system indices rank owns: contiguous blocks, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError('rank %d outside world of %d' % (rank, world))
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def weak_shard(per_rank, world, rank):
    """weak scaling: every rank owns `per_rank` systems; returns its [start, stop) in the global list."""
    return rank * per_rank, (rank + 1) * per_rank


def system_seed(base_seed, global_index):
    """thermostat seed of a system (main.cpp:459); upside_hip_init_md adds the LOCAL index itself, so a rank
    passes system_seed(base, start_of_its_shard)."""
    return (base_seed + global_index) & 0xFFFFFFFF


def barrier(dist, sync_device=None):
    """device sync + barrier + device sync, as the bench contract brackets the timed region"""
    if sync_device is not None:
        sync_device()
    if dist is not None and dist.is_initialized():
        dist.barrier()
    if sync_device is not None:
        sync_device()


def max_over_ranks(dist, seconds, device='cpu'):
    import torch
    if dist is None or not dist.is_initialized():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device='cpu'):
    import torch
    if dist is None or not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def job_throughput(dist, units_this_rank, seconds_this_rank, device='cpu'):
    """whole-job rate: units all ranks processed / slowest rank's time"""
    total = sum_over_ranks(dist, units_this_rank, device)
    worst = max_over_ranks(dist, seconds_this_rank, device)
    return total / worst, worst
