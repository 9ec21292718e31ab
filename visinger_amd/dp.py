"""Data-parallel plumbing of the synthesis path: one process per GPU, utterances sharded by the reference's strided
rule, no data-path collective.  (The only collectives are the barrier and the max-over-ranks of the step time.)

Reference: tasks/base.py:130-133  ``batches = [b[rank::world] for b in batches if len(b) % world == 0]``.
"""
import torch


def shard_batch(tensors, rank, world):
    """Strided utterance shard of a global batch (every tensor's dim 0 is the utterance).  Batches whose size is
    not divisible by the world size are dropped by the reference; here that is an error."""
    out = []
    for t in tensors:
        if t.shape[0] % world != 0:
            raise ValueError(f"global batch {t.shape[0]} is not divisible by world size {world} (tasks/base.py:133)")
        out.append(t[rank::world].contiguous())
    return out


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (step time) over the default process group (RCCL on GPUs, gloo on CPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
