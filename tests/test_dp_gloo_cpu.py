"""world_size-2 gloo test (CPU) of the data-parallel plumbing bench.py uses for --gpus N: the reference's strided
utterance shard (tasks/base.py:130-133) partitions the global batch exactly, and the step time is the max over ranks."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visinger_amd.dp import shard_batch, max_over_ranks
    sys.path.insert(0, ROOT)
    import bench
    gb = bench.synthetic_batch(4 * world, 64, 8, 64, 1234, "cpu")     # every rank builds the same global batch
    shard = shard_batch(gb, rank, world)
    # gather the shards on every rank and rebuild the global batch
    for full, part in zip(gb, shard):
        parts = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(parts, part)
        rebuilt = torch.empty_like(full)
        for r in range(world):
            rebuilt[r::world] = parts[r]
        assert torch.equal(rebuilt, full)
        assert part.shape[0] == full.shape[0] // world
    t = max_over_ranks(1.0 + rank)
    ok = (t == float(world))
    try:
        shard_batch([torch.zeros(3, 2)], rank, world)
        ok = False
    except ValueError:
        pass
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_strided_shard_and_max_time_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
