"""Temperature replica exchange over the GPUs of one node (BASELINE.json configs[3]; main.cpp:616-667 loop with
ReplicaExchange::attempt_swaps every --replica-interval).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        tools/remd.py --fixture proteinG56_7A --temperatures 64 --t-low 0.7 --t-high 1.0 --rounds 600

Rank r owns a contiguous block of the temperature ladder (most neighbour swaps stay on one GPU).  Per attempt the
ranks all-gather one energy per replica over RCCL, every rank reaches the identical Metropolis verdicts and the coordinates
of accepted cross-GPU pairs move point to point; nothing else crosses xGMI.  Also runs as a single process.

--exchange rccl (default): the exchange runs inside the library (upside_hip_comm_*, comm_rccl.cpp: ncclAllGather, verdicts on
the device, ncclSend/ncclRecv), enqueued behind the MD steps with no host staging; the verdicts are read back only for the
statistics printed here.  --exchange host: the same protocol driven from Python over torch.distributed
(replicas.exchange_swap_set; gloo-testable, tests/test_replicas_gloo.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fixture', default='proteinG56_7A')
    ap.add_argument('--temperatures', type=int, default=8)
    ap.add_argument('--t-low', type=float, default=0.7)
    ap.add_argument('--t-high', type=float, default=1.0)
    ap.add_argument('--rounds', type=int, default=300, help='integration cycles (3 MD steps each)')
    ap.add_argument('--replica-interval', type=int, default=5, help='rounds between exchange attempts (README.md:189-193)')
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--exchange', choices=['rccl', 'host'], default='rccl')
    args = ap.parse_args()

    pkg = load_package()
    rep = pkg.replicas
    rank, local_rank, world = rep.world_from_env()
    import torch
    dist = None
    device = 'cpu'
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        device = torch.device('cuda', local_rank)
    if args.temperatures % world:
        raise SystemExit('--temperatures must be a multiple of the number of ranks')
    per_rank = args.temperatures // world
    lo, hi = rep.weak_shard(per_rank, world, rank)
    ladder = rep.geometric_ladder(args.t_low, args.t_high, args.temperatures)
    beta = (1.0 / ladder).astype('f4')
    fixture = os.path.join(ROOT, 'tests', 'golden', args.fixture + '.up')
    ens = pkg.engine.Ensemble(fixture, per_rank, device=local_rank)
    ens.set_pos(pkg.config.read_pos(fixture))
    ens.init_md(ladder[lo:hi], rep.system_seed(args.seed, lo))
    swap_sets = rep.neighbour_swap_sets(args.temperatures)
    if args.exchange == 'rccl':
        uid = [ens.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        ens.comm_init(rank, world, uid[0], ladder)
    n_attempt = np.zeros(len(swap_sets), 'i8'); n_success = np.zeros(len(swap_sets), 'i8')
    replica_index = np.arange(args.temperatures)            # which starting replica sits in each temperature slot
    rep.barrier(dist, torch.cuda.synchronize)
    t0 = time.perf_counter()
    done = 0
    while done < args.rounds:
        n = min(args.replica_interval, args.rounds - done)
        ens.run_rounds(n)
        done += n
        draw = 0
        if args.exchange == 'host':
            energy = rep.all_gather_f32(dist, ens.energies(), device)      # one force evaluation per attempt ...
        for k, pairs in enumerate(swap_sets):
            if args.exchange == 'rccl':
                acc = ens.comm_replica_swap(pairs, args.seed, done, k == 0, want_accepted=True)
            else:
                acc, draw = rep.exchange_swap_set(dist, ens, pairs, beta, args.seed, done, draw, device, energy_global=energy)
                energy = rep.swap_energies(energy, pairs, acc)               # ... the later sets see the traded energies
            n_attempt[k] += len(pairs); n_success[k] += int(acc.sum())
            for (s1, s2), ok in zip(pairs, acc):
                if ok:
                    replica_index[[s1, s2]] = replica_index[[s2, s1]]
    rep.barrier(dist, torch.cuda.synchronize)
    elapsed = rep.max_over_ranks(dist, time.perf_counter() - t0, device)
    energy = rep.all_gather_f32(dist, ens.energies(), device)
    if rank == 0:
        print(json.dumps(dict(fixture=args.fixture, n_gpus=world, temperatures=[float(t) for t in ladder],
                              md_steps_per_replica=3 * args.rounds, seconds=elapsed,
                              system_steps_per_s=3 * args.rounds * args.temperatures / elapsed,
                              swap_acceptance=[float(s) / max(a, 1) for s, a in zip(n_success, n_attempt)],
                              replica_index=replica_index.tolist(), final_energy=[float(e) for e in energy])), flush=True)
    ens.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    main()
