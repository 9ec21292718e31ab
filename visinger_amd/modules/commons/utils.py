"""Small helpers of the hot path -- same names / semantics as the reference's modules/commons/utils.py:71-110."""
import torch
import torch.nn as nn


def Embedding(num_embeddings, embedding_dim, padding_idx=None):
    """modules/commons/utils.py:71-76"""
    m = nn.Embedding(num_embeddings, embedding_dim, padding_idx=padding_idx)
    nn.init.normal_(m.weight, mean=0.0, std=embedding_dim ** -0.5)
    if padding_idx is not None:
        nn.init.constant_(m.weight[padding_idx], 0)
    return m


def sequence_mask(length, max_length=None):
    """modules/commons/utils.py:79-83"""
    if max_length is None:
        max_length = length.max()
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


def slice_segments(x, ids_str, segment_size=4):
    """modules/commons/utils.py:86-92 -- one launch (vs_slice_segments) instead of a Python loop over the batch."""
    if torch.is_grad_enabled() and x.requires_grad:      # training: differentiable gather (PyTorch-ROCm), same indexing
        idx = ids_str.to(device=x.device, dtype=torch.long)[:, None] + torch.arange(segment_size, device=x.device)[None, :]
        return torch.gather(x, 2, idx[:, None, :].expand(-1, x.size(1), -1))
    from ...ops import slice_segments as _slice_segments_hip
    return _slice_segments_hip(x, ids_str, segment_size)


def rand_slice_segments(x, segment_size=4):
    """modules/commons/utils.py:95-100.  The start ids come from the CPU generator (torch.rand([batch]) then
    .to(device)), exactly like the reference, so they are bit-reproducible across devices."""
    batch, _, t_len = x.size()
    ids_str_max = t_len - segment_size + 1
    ids_str = (torch.rand([batch]).to(device=x.device) * ids_str_max).to(dtype=torch.long)
    return slice_segments(x, ids_str, segment_size), ids_str


def init_weights(m, mean=0.0, std=0.01):
    """modules/commons/utils.py:103-106"""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        m.weight.data.normal_(mean, std)


def get_padding(kernel_size, dilation=1):
    """modules/commons/utils.py:109-110"""
    return int((kernel_size * dilation - dilation) / 2)
