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


# ---- replica exchange across ranks (main.cpp:227-275) -------------------------------------------
def all_gather_f32(dist, local, device='cpu'):
    """concatenate one float32 vector per rank in rank order (the ncclAllGather of one energy per replica)"""
    import numpy as np
    import torch
    local = np.ascontiguousarray(local, dtype='f4')
    if dist is None or not dist.is_initialized():
        return local.copy()
    t = torch.from_numpy(local).to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out).cpu().numpy()


def exchange_swap_set(dist, ensemble, pairs, beta_global, base_seed, round_num, draw0=0, device='cpu', decide=None,
                      energy_global=None):
    """One swap set of a replica-exchange attempt over systems spread across ranks (weak shard: rank r owns the
    global systems [r*n, (r+1)*n), n = ensemble.n_system).

    1. every rank evaluates the energies of its systems and all-gathers them (one float per replica);
    2. every rank runs the identical Metropolis test on the identical arrays (`upside_replica_decide`: shared
       counter RNG, a uniform drawn only for a rejectable pair) -- no verdict is communicated;
    3. accepted pairs exchange coordinates: on-rank pairs on the device, cross-rank pairs by one grouped
       send/recv of 3*n_atom floats per pair.  Momenta and temperatures stay where they are (main.cpp:244-247).

    `ensemble` needs n_system, energies(), get_system_pos(i), set_system_pos(i, x), swap_systems(i, j).
    `energy_global`: the gathered energies if the caller already has them (in a temperature-only exchange the later
    swap sets of one attempt need no new force evaluation: accepted pairs just trade their energies, `swap_energies`).
    Returns (accepted bool array over `pairs`, next draw index)."""
    import numpy as np
    import torch
    if decide is None:
        from .engine import replica_decide as decide
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    n = ensemble.n_system
    lo = rank * n
    pairs = np.asarray(pairs, dtype='i4').reshape(-1, 2)
    energy = all_gather_f32(dist, ensemble.energies(), device) if energy_global is None else np.asarray(energy_global, 'f4')
    accepted, draw = decide(pairs, beta_global, energy, base_seed, round_num, draw0)
    ops, incoming, local = [], [], []
    for (s1, s2), ok in zip(pairs.tolist(), accepted.tolist()):
        if not ok:
            continue
        r1, r2 = s1 // n, s2 // n
        if r1 == rank and r2 == rank:
            local.append((s1 - lo, s2 - lo))
        elif rank in (r1, r2):
            mine, peer = (s1, r2) if r1 == rank else (s2, r1)
            out = torch.from_numpy(np.ascontiguousarray(ensemble.get_system_pos(mine - lo))).to(device)
            inc = torch.empty_like(out)
            ops.append(dist.P2POp(dist.isend, out, peer))
            ops.append(dist.P2POp(dist.irecv, inc, peer))
            incoming.append((mine - lo, inc))
    if local:       # all on-rank pairs of the set in one device launch
        if hasattr(ensemble, 'swap_system_pairs'):
            ensemble.swap_system_pairs(local)
        else:
            for a, b in local:
                ensemble.swap_systems(a, b)
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for sys_local, inc in incoming:
            ensemble.set_system_pos(sys_local, inc.cpu().numpy())
    return accepted, draw


def swap_energies(energy_global, pairs, accepted):
    """energies after a swap set of a temperature-only exchange: the coordinates of an accepted pair changed places,
    so did their (Hamiltonian-independent) energies"""
    import numpy as np
    e = np.array(energy_global, dtype='f4', copy=True)
    for (s1, s2), ok in zip(np.asarray(pairs).reshape(-1, 2).tolist(), np.asarray(accepted).tolist()):
        if ok:
            e[s1], e[s2] = e[s2], e[s1]
    return e


def geometric_ladder(t_low, t_high, n):
    """temperature ladder of a replica-exchange run (README.md:189-193 pattern)"""
    import numpy as np
    return (t_low * (t_high / t_low) ** (np.arange(n) / max(n - 1, 1))).astype('f4')


def neighbour_swap_sets(n):
    """the two alternating swap sets of nearest temperature neighbours: (0,1),(2,3),... and (1,2),(3,4),..."""
    return [[(i, i + 1) for i in range(0, n - 1, 2)], [(i, i + 1) for i in range(1, n - 1, 2)]]
