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
