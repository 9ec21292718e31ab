"""models/commons/align_ops.py:22-26 of the reference: frame expansion by the 1-based mel2ph index."""
from ...ops import expand_states as _expand_states_hip


def expand_states(h, mel2token):
    """h: [B, T_ph, H]; mel2token: int64 [B, T_mel] with 0 = padding -> [B, T_mel, H] (index 0 reads the zero pad row
    the reference prepends).  Integer indexing on the GPU (vs_expand_states): values are moved, never recomputed."""
    return _expand_states_hip(h, mel2token)
