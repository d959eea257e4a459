"""N>1 path of bench.py (replicas only) on CPU: world_size-2 gloo processes check that the
shards are disjoint and cover the global system list, that seeds follow the global system index, and that the
job throughput is (units of all ranks) / (slowest rank's time)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per_rank, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    rep = load_package().replicas
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        assert rep.world_from_env() == (rank, rank, world)
        # weak scaling shard used by bench.py and the strong (fixed total) split
        lo, hi = rep.weak_shard(per_rank, world, rank)
        mine = torch.zeros(world * per_rank, dtype=torch.int64)
        mine[lo:hi] = 1
        dist.all_reduce(mine)
        assert bool((mine == 1).all()), 'weak shards must tile the global list exactly once'
        total = 7 * world + 3
        lo2, hi2 = rep.shard(total, world, rank)
        cover = torch.zeros(total, dtype=torch.int64)
        cover[lo2:hi2] = 1
        dist.all_reduce(cover)
        assert bool((cover == 1).all())
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([hi2 - lo2]))
        assert max(int(x) for x in sizes) - min(int(x) for x in sizes) <= 1
        # seeds: global index, so the union over ranks equals a single-process run of world*per_rank systems
        seeds = torch.tensor([rep.system_seed(1000, lo) + i for i in range(per_rank)], dtype=torch.int64)
        allseeds = [torch.zeros(per_rank, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allseeds, seeds)
        assert torch.cat(allseeds).tolist() == [1000 + i for i in range(world * per_rank)]
        # timing: the slower rank decides
        rep.barrier(dist)
        seconds = 2.0 if rank == 1 else 1.0
        value, worst = rep.job_throughput(dist, per_rank * 30, seconds)
        assert worst == 2.0
        assert abs(value - world * per_rank * 30 / 2.0) < 1e-9
        with open(os.path.join(out_dir, 'ok%d' % rank), 'w') as f:
            f.write('%r' % value)
    finally:
        dist.destroy_process_group()


def test_two_rank_replica_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 8, str(tmp_path)), nprocs=world, join=True)
    vals = [open(os.path.join(str(tmp_path), 'ok%d' % r)).read() for r in range(world)]
    assert vals[0] == vals[1]


def test_single_process_degenerates():
    from __graft_entry__ import load_package
    rep = load_package().replicas
    assert rep.shard(10, 1, 0) == (0, 10)
    assert rep.weak_shard(64, 4, 3) == (192, 256)
    assert rep.max_over_ranks(None, 1.5) == 1.5
    value, worst = rep.job_throughput(None, 300, 2.0)
    assert value == 150.0 and worst == 2.0
    with pytest.raises(ValueError):
        rep.shard(10, 2, 2)


# ---- replica exchange (main.cpp:227-275) -------------------------------------------------------------------
import ctypes as ct  # noqa: E402
import numpy as np  # noqa: E402


def _expected_decisions(pairs, beta, energy, seed, round_num, draw0=0):
    """restatement of main.cpp:251-273 for temperature exchange with the oracle's RandomGenerator
    (oracle_random_uniform4 is pinned to the Random123 known answers in test_oracle_pinning.py)"""
    import parity_util as P
    orc = P.oracle_library().calc
    orc.oracle_random_uniform4.restype = None
    orc.oracle_random_uniform4.argtypes = [ct.c_void_p, ct.c_uint32, ct.c_uint32, ct.c_uint32, ct.c_uint64, ct.c_uint32]
    beta = np.asarray(beta, 'f4'); energy = np.asarray(energy, 'f4')
    acc, draw = [], draw0
    for s1, s2 in pairs:
        new = np.float32(-beta[s1] * energy[s2]) + np.float32(-beta[s2] * energy[s1])
        old = np.float32(-beta[s1] * energy[s1]) + np.float32(-beta[s2] * energy[s2])
        lb = np.float32(new - old)
        ok = True
        if lb < 0:
            u = np.zeros(4, 'f4')
            orc.oracle_random_uniform4(u.ctypes.data, seed, 1, 0, round_num, draw)   # stream 1 = REPLICA_EXCHANGE
            draw += 1
            if np.float32(np.exp(lb, dtype='f4')) < u[0]:
                ok = False
        acc.append(ok)
    return np.array(acc), draw


def test_replica_decide_matches_reference_rule():
    from __graft_entry__ import load_package
    pkg = load_package()
    rs = np.random.RandomState(3)
    n = 16
    temps = pkg.replicas.geometric_ladder(0.7, 1.1, n)
    beta = (1.0 / temps).astype('f4')
    n_reject = 0
    for round_num in range(40):
        energy = (-300 + 4.0 * rs.randn(n)).astype('f4')
        draw = 0
        for pairs in pkg.replicas.neighbour_swap_sets(n):
            got, draw_got = pkg.engine.replica_decide(pairs, beta, energy, 77, round_num, draw)
            want, draw_want = _expected_decisions(pairs, beta, energy, 77, round_num, draw)
            assert np.array_equal(got, want) and draw_got == draw_want
            n_reject += int((~got).sum())
            draw = draw_got
    assert n_reject > 20          # the random branch was exercised
    assert pkg.replicas.neighbour_swap_sets(5) == [[(0, 1), (2, 3)], [(1, 2), (3, 4)]]


class _FakeEnsemble(object):
    """numpy stand-in with the interface exchange_swap_set needs; energy = sum of the coordinates"""

    def __init__(self, pos):
        self.pos = pos.copy(); self.n_system = pos.shape[0]

    def energies(self):
        return self.pos.sum(axis=(1, 2)).astype('f4')

    def get_system_pos(self, i):
        return self.pos[i].copy()

    def set_system_pos(self, i, x):
        self.pos[i] = x

    def swap_systems(self, i, j):
        self.pos[[i, j]] = self.pos[[j, i]]


def _exchange_worker(rank, world, port, out_dir, per_rank=4, n_round=6):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    pkg = load_package()
    rep = pkg.replicas
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n_atom = 6
        n = per_rank * world
        rs = np.random.RandomState(11)
        all_pos = rs.randn(n, n_atom, 3).astype('f4')
        beta = (1.0 / rep.geometric_ladder(0.5, 2.0, n)).astype('f4')
        lo, hi = rep.weak_shard(per_rank, world, rank)
        ens = _FakeEnsemble(all_pos[lo:hi])
        ens2 = _FakeEnsemble(all_pos[lo:hi])                 # one energy evaluation per attempt, energies traded between sets
        ref = _FakeEnsemble(all_pos)                         # the same run in one process
        n_cross = 0
        for round_num in range(n_round):
            draw = draw_ref = draw2 = 0
            energy = rep.all_gather_f32(dist, ens2.energies())
            for pairs in rep.neighbour_swap_sets(n):         # the second set has the cross-rank pairs (per_rank - 1, per_rank), ...
                acc, draw = rep.exchange_swap_set(dist, ens, pairs, beta, 5, round_num, draw)
                n_cross += sum(1 for (a, b), ok in zip(pairs, acc) if ok and a // per_rank != b // per_rank)
                acc_ref, draw_ref = rep.exchange_swap_set(None, ref, pairs, beta, 5, round_num, draw_ref)
                acc2, draw2 = rep.exchange_swap_set(dist, ens2, pairs, beta, 5, round_num, draw2, energy_global=energy)
                energy = rep.swap_energies(energy, pairs, acc2)
                assert np.array_equal(acc, acc_ref) and draw == draw_ref
                assert np.array_equal(acc2, acc_ref) and draw2 == draw_ref
                assert np.array_equal(energy, ref.energies())
            assert np.array_equal(ens.pos, ref.pos[lo:hi]), 'sharded exchange must equal the single-process one'
            assert np.array_equal(ens2.pos, ref.pos[lo:hi])
        moved = not np.array_equal(ref.pos, all_pos)
        with open(os.path.join(out_dir, 'x%d' % rank), 'w') as f:
            f.write('%d' % moved)
        with open(os.path.join(out_dir, 'c%d' % rank), 'w') as f:
            f.write('%d' % n_cross)
    finally:
        dist.destroy_process_group()


def test_two_rank_replica_exchange(tmp_path):
    world = 2
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert [open(os.path.join(str(tmp_path), 'x%d' % r)).read() for r in range(world)] == ['1', '1']


def test_eight_rank_remd64_plan_equals_single_process(tmp_path):
    """BASELINE configs[3] as it is placed on an 8-GPU node: 64 temperatures, 8 per rank, so the second swap set of
    every attempt has 7 pairs that straddle a rank boundary.  Eight gloo ranks must reproduce the single-process exchange frame for
    frame -- verdicts, generator position, coordinates after every set, and the traded energies of the one-evaluation form -- and
    accepted cross-rank swaps must actually occur."""
    world, per_rank = 8, 8
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path), per_rank, 8), nprocs=world, join=True)
    assert [open(os.path.join(str(tmp_path), 'x%d' % r)).read() for r in range(world)] == ['1'] * world
    crossed = [int(open(os.path.join(str(tmp_path), 'c%d' % r)).read()) for r in range(world)]
    assert len(set(crossed)) == 1 and crossed[0] > 0, crossed      # every rank saw the same accepted cross-rank swaps, and there were some
    from __graft_entry__ import load_package
    rep = load_package().replicas
    sets = rep.neighbour_swap_sets(world * per_rank)
    assert sum(1 for a, b in sets[0] if a // per_rank != b // per_rank) == 0
    assert sum(1 for a, b in sets[1] if a // per_rank != b // per_rank) == world - 1
