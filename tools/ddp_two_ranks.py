#!/usr/bin/env python3
"""One rank of a two-rank DistributedDataParallel(VISingerTrainer) step on ONE GPU (gloo rendezvous, both ranks on cuda:0) -- the product's
real training path with world > 1 (reference: utils/commons/trainer.py:117-138,497-507 wraps the task in DDP; tasks/base.py:130-133 shards
the batch by rank).  Started as a FRESH child process by tests/test_ddp_two_ranks_gpu.py (RANK / WORLD_SIZE / MASTER_* in the environment).

Every rank builds the same trainer (same seed) and the same global batch, takes its strided shard, and runs both optimizer passes under
DDP; rank 0 also runs the same passes on an identical, unwrapped trainer over the GLOBAL batch and compares, after each backward, every
gradient of the network being trained (all-reduced = averaged over the ranks) with the single-process gradient -- then prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from visinger_amd.dp import shard_batch
    from visinger_amd.models.visinger import hop256_hparams
    from visinger_amd.train import VISingerTrainer, synthetic_train_batch
    # the reference architecture at a reduced width (the discriminators at full size), no dropout: the two runs must see the same numbers
    hp = hop256_hparams(p_dropout=0.0, hidden_size=64, ffn_filter_channels=128, gin_channels=32, enc_layers=2, pitch_predictor_layers=2,
                        frame_prior_layers=2, phoneme_predictor_layers=1, initial_upsample_channels=128, segment_size=8, num_linear_bins=65)

    def make():
        torch.manual_seed(77)
        return VISingerTrainer(64, 117, 131, hp).to(dev).configure().train()

    tr = make()
    ddp = torch.nn.parallel.DistributedDataParallel(tr, device_ids=[0], find_unused_parameters=True)
    B, T = 4, 64
    gb = synthetic_train_batch(B, T, T // 8, tr.hop, 64, hp["num_linear_bins"], 1234, dev)
    # the f0 loss divides by the number of VOICED frames of the batch it sees (tasks/visinger.py:139-142): with per-item voicing patterns the
    # mean over two shards is not the global batch's value (in the reference's own DDP run as well) -- one pattern for every item makes the
    # sharded and the global step the same function
    gb["uv"] = gb["uv"][:1].repeat(B, 1).contiguous()
    g = torch.Generator().manual_seed(5)
    gb["noise_q"] = torch.randn(B, hp["hidden_size"], T, generator=g).to(dev)       # injected draws: posterior noise, segment starts
    gb["u_slice"] = torch.rand(B, generator=g).to(dev)
    keys = list(gb)
    mine = dict(zip(keys, shard_batch([gb[k] for k in keys], rank, world)))
    ref = make() if rank == 0 else None
    report = {"world": world, "passes": []}
    for opt_idx, opt_name in ((0, "opt_gen"), (1, "opt_disc")):
        tr.backward_pass(mine, opt_idx, runner=ddp)
        own = tr.model if opt_idx == 0 else tr.mel_disc
        if rank == 0:
            ref.backward_pass(gb, opt_idx)
            own_ref = ref.model if opt_idx == 0 else ref.mel_disc
            worst, n, missing, offenders = 0.0, 0, 0, []
            gmax = max(float(q.grad.abs().max()) for q in own_ref.parameters() if q.grad is not None)
            for (name, p), (_, q) in zip(own.named_parameters(), own_ref.named_parameters()):
                if (p.grad is None) != (q.grad is None):
                    missing += 1
                    continue
                if q.grad is None:
                    continue
                n += 1
                # relative to the parameter's own largest gradient, floored at 1e-6 of the network's largest (a gradient that is zero up
                # to rounding -- e.g. a bias in front of a normalisation over the same axis -- has no scale of its own)
                scale = max(float(q.grad.abs().max()), 1e-6 * gmax)
                e = float((p.grad - q.grad).abs().max()) / scale
                offenders.append((e, name, float(q.grad.abs().max())))
                worst = max(worst, e)
            offenders.sort(reverse=True)
            other = tr.mel_disc if opt_idx == 0 else tr.model
            report["passes"].append({"optimizer": opt_name, "gradients_compared": n, "presence_mismatches": missing,
                                     "worst_rel_err": worst, "largest_gradient": gmax, "worst_parameters": [(nm, e, g_) for e, nm, g_ in offenders[:4]],
                                     "other_network_has_grads": any(p.grad is not None for p in other.parameters())})
        for t in ([tr] if rank else [tr, ref]):           # the same optimizer step on both, so that the second pass starts from equal weights
            getattr(t, opt_name).step()
            getattr(t, opt_name).zero_grad(set_to_none=True)
    # after both steps every rank must hold the same weights (DDP's invariant)
    flat = torch.cat([p.detach().flatten() for p in tr.parameters()])
    ck = torch.stack([flat.double().sum(), flat.double().abs().sum()]).cpu()
    gathered = [torch.zeros_like(ck) for _ in range(world)]
    dist.all_gather(gathered, ck)
    if rank == 0:
        report["weights_equal_across_ranks"] = bool(all(torch.equal(gathered[0], t) for t in gathered))
        report["native_library"] = os.path.basename(__import__("visinger_amd._lib", fromlist=["LIB_PATH"]).LIB_PATH)
        print(json.dumps(report), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
