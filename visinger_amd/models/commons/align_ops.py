"""models/commons/align_ops.py:22-26 of the reference: frame expansion by the 1-based mel2ph index."""
from ...ops import expand_states as _expand_states_hip
from ...ops import mel2token_to_dur as _mel2token_to_dur_hip


def expand_states(h, mel2token):
    """h: [B, T_ph, H]; mel2token: int64 [B, T_mel] with 0 = padding -> [B, T_mel, H] (index 0 reads the zero pad row
    the reference prepends).  Integer indexing on the GPU (vs_expand_states): values are moved, never recomputed."""
    return _expand_states_hip(h, mel2token)


def clip_mel2token_to_multiple(mel2token, frames_multiple):
    """models/commons/align_ops.py:16-19"""
    max_frames = mel2token.shape[1] // frames_multiple * frames_multiple
    return mel2token[:, :max_frames]


def mel2token_to_dur(mel2token, T_txt=None, max_dur=None):
    """utils/audio/align.py:105-129 for device tensors: frames per token from the 1-based alignment (0 = padding), with the
    reference's handling of a missing batch dim and of ``T_txt=None`` (largest index present)."""
    has_batch = mel2token.dim() == 2
    m = mel2token if has_batch else mel2token[None]
    if T_txt is None:
        T_txt = int(m.max())
    dur = _mel2token_to_dur_hip(m, T_txt, max_dur)
    return dur if has_batch else dur[0]
