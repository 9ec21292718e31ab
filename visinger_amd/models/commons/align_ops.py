"""models/commons/align_ops.py:22-26 of the reference: frame expansion by the 1-based mel2ph index."""
import torch
import torch.nn.functional as F


def expand_states(h, mel2token):
    """h: [B, T_ph, H]; mel2token: int64 [B, T_mel] with 0 = padding -> [B, T_mel, H] (row 0 is the zero pad row).
    Integer indexing: bit-exact by construction (a gather moves values, no arithmetic)."""
    h = F.pad(h, [0, 0, 1, 0])
    mel2token_ = mel2token[..., None].expand(-1, -1, h.shape[-1])
    return torch.gather(h, 1, mel2token_)
