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


def _shard8_worker(rank, world, port, q):
    """BASELINE configs[3]: 256 utterances over 8 ranks = 8 x 32, the reference's strided shard batch[rank::8] (tasks/base.py:130-133)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visinger_amd.dp import shard_batch, max_over_ranks
    g = torch.Generator().manual_seed(1234)                  # every rank builds the same global batch (ids, lengths, tokens)
    ids = torch.arange(256)
    lens = torch.randint(512, 1025, (256,), generator=g)
    tokens = torch.randint(4, 64, (256, 128), generator=g)
    my_ids, my_lens, my_tok = shard_batch([ids, lens, tokens], rank, world)
    ok = my_ids.shape[0] == 32 and torch.equal(my_ids, torch.arange(rank, 256, world)) and torch.equal(my_tok, tokens[rank::world])
    got = [torch.empty_like(my_ids) for _ in range(world)]
    dist.all_gather(got, my_ids)                             # (a test-side gather: the data path itself has no collective)
    ok = ok and torch.equal(torch.sort(torch.cat(got)).values, ids)            # a partition: every utterance on exactly one rank
    frames = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(frames, my_lens.sum().view(1))
    ok = ok and int(torch.stack(frames).sum()) == int(lens.sum())
    ok = ok and max_over_ranks(10.0 + rank) == 10.0 + world - 1               # the step time is the slowest rank's
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_config4_strided_shard_world8():
    """B = 256 -> 8 x 32 over gloo, world_size 8: the partition bench.py --gpus 8 uses, on CPU"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def _ddp_worker(rank, world, port, q):
    """Training-path collective: the discriminator loss of visinger_amd.train under stock DDP over gloo.
    Each rank sees its strided shard; the all-reduced gradient must equal the single-process gradient on the global
    batch.  The product discriminators run on HIP kernels only (no CPU path), so a small PyTorch network with the same
    calling convention -- (real, generated) -> (logits_real, logits_generated, fmaps, fmaps) -- stands in for them here:
    what is under test is the shard + all-reduce plumbing and the loss, not the convolutions."""
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from torch.nn.parallel import DistributedDataParallel as DDP
        from visinger_amd.dp import shard_batch
        from visinger_amd.train import discriminator_loss

        class StandInDiscriminators(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.nets = torch.nn.ModuleList(
                    torch.nn.Sequential(torch.nn.Conv1d(1, 8, 15, s, padding=7), torch.nn.LeakyReLU(0.1),
                                        torch.nn.Conv1d(8, 1, 3, 1, padding=1)) for s in (1, 2, 3))

            def forward(self, y, y_hat):
                return [n(y).flatten(1) for n in self.nets], [n(y_hat).flatten(1) for n in self.nets], [], []

        torch.manual_seed(7)                                   # same weights on every rank
        disc = StandInDiscriminators()
        g = torch.Generator().manual_seed(11)
        real = torch.rand(2 * world, 1, 1200, generator=g) - 0.5
        fake = 0.3 * torch.randn(2 * world, 1, 1200, generator=g)

        def grads(net, real, fake):
            disc.zero_grad(set_to_none=True)
            d_tgt, d_gen, _, _ = net(real, fake)
            loss = discriminator_loss(d_tgt, d_gen)
            loss.backward()
            return torch.cat([p.grad.flatten() for p in disc.parameters()]).clone(), float(loss)

        full, loss_full = grads(disc, real, fake)
        ddp = DDP(disc, find_unused_parameters=True)
        r, f = shard_batch([real, fake], rank, world)
        mine, loss_mine = grads(ddp, r, f)
        losses = [None] * world
        dist.all_gather_object(losses, loss_mine)
        scale = float(full.abs().max())
        ok = float((mine - full).abs().max()) <= 2e-5 * scale and abs(sum(losses) / world - loss_full) <= 1e-5 * abs(loss_full)
        q.put((rank, bool(ok)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:   # surface the failure instead of letting the parent wait for the queue
        q.put((rank, repr(e)))
        raise


def test_ddp_gradient_allreduce_matches_global_batch_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment starts two rank processes itself (the reference spawns its ranks
    too: utils/commons/trainer.py:117-138); --dry-run + VS_BENCH_BACKEND=gloo rehearses exactly that launch path on CPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["VS_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks"] == [0, 1] and line["processes"] == 2 and line["max_over_ranks"] == 2.0
    assert line["launcher_pid_is_a_rank"] is False           # the launcher only spawns: it never joins the group or touches a GPU
    # a world size that is not --gpus is refused (non-zero exit, no line), e.g. a stale torchrun environment
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="1"),
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_bench_launcher_stops_all_ranks_when_one_dies():
    """spawn_ranks waited for its children one after the other -- a rank that died left its siblings in dist.barrier()
    (and the launcher) hanging.  Now the launcher polls all of them, terminates the survivors and exits non-zero."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(VS_BENCH_BACKEND="gloo", VS_BENCH_DIE_RANK="1")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 7, (out.returncode, out.stderr[-2000:])
    assert "stopping the other 1 rank(s)" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]          # no line for a broken run
    assert time.time() - t0 < 240
