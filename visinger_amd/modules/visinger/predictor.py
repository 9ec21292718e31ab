"""Frame-level predictors of VISinger on the MI355X-native transformer: the pitch predictor (log-f0 + voicing logits per
frame) and the phoneme predictor (CTC log-probabilities per frame).  Drop-in for the reference's
``modules/visinger/predictor.py`` (PitchPredictor :7-19, PhonemePredictor :22-35): same constructor arguments, same
sub-module names -- ``pitch_predictor`` / ``linear`` and ``phoneme_predictor`` / ``ph_proj`` -- hence the same state-dict keys.

Both are "relative-attention encoder + 1x1 projection"; the shared plumbing lives in ``_EncoderHead``.  The projections are
1x1 convs with 2 and ``dict_size`` output rows: the 2-row one takes the engine's small-C_out VALU path
(csrc/conv_engine.hip, conv_small_kernel), the other one the MFMA path.
"""
import torch
import torch.nn as nn

from ..hipconv import HipConv1d
from ..rel_transformer import RelativeEncoder


class _EncoderHead(nn.Module):
    """RelativeEncoder -> HipConv1d(width, n_out, 1), registered under the attribute names given by the subclass."""

    ENCODER = HEAD = None      # attribute (= state-dict prefix) names, set by the subclasses

    def _assemble(self, width, n_out, encoder_kwargs):
        self.add_module(self.ENCODER, RelativeEncoder(width, **encoder_kwargs))
        self.add_module(self.HEAD, HipConv1d(width, n_out, 1))

    def _project(self, frames, frame_mask, cond=None):
        hidden = getattr(self, self.ENCODER)(frames, frame_mask, g=cond)
        return getattr(self, self.HEAD)(hidden)                       # [B, n_out, T]


class PitchPredictor(_EncoderHead):
    ENCODER, HEAD = "pitch_predictor", "linear"

    def __init__(self, in_dim, filter_channels, n_heads, n_layers, kernel_size, p_dropout, gin_channels, out_dim=2):
        super().__init__()
        self._assemble(in_dim, out_dim, dict(filter_channels=filter_channels, n_heads=n_heads, n_layers=n_layers,
                                             kernel_size=kernel_size, p_dropout=p_dropout, gin_channels=gin_channels))

    def forward(self, x, x_mask, spk_emb):
        """x [B, in_dim, T], x_mask [B, 1, T], spk_emb [B, gin, 1] -> [B, T, out_dim] (channel 0: log-f0, 1: voicing logit)"""
        return self._project(x, x_mask, spk_emb).permute(0, 2, 1)


class PhonemePredictor(_EncoderHead):
    ENCODER, HEAD = "phoneme_predictor", "ph_proj"

    def __init__(self, dict_size, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout):
        super().__init__()
        self._assemble(hidden_channels, dict_size, dict(filter_channels=filter_channels, n_heads=n_heads, n_layers=n_layers,
                                                        kernel_size=kernel_size, p_dropout=p_dropout))

    def forward(self, x, x_mask):
        """x [B, hidden, T] -> log-probabilities over the phoneme dictionary, [B, dict_size, T]"""
        return torch.log_softmax(self._project(x, x_mask), dim=1)
