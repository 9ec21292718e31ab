"""MI355X-native mirror of the reference's modules/visinger/predictor.py:7-35."""
import torch.nn as nn
import torch.nn.functional as F

from ..hipconv import HipConv1d
from ..rel_transformer import RelativeEncoder


class PitchPredictor(nn.Module):
    """predictor.py:7-19"""

    def __init__(self, in_dim, filter_channels, n_heads, n_layers, kernel_size, p_dropout, gin_channels, out_dim=2):
        super().__init__()
        self.pitch_predictor = RelativeEncoder(in_dim, filter_channels, n_heads, n_layers=n_layers,
                                               gin_channels=gin_channels, kernel_size=kernel_size, p_dropout=p_dropout)
        self.linear = HipConv1d(in_dim, out_dim, 1)

    def forward(self, x, x_mask, spk_emb):
        x = self.pitch_predictor(x, x_mask, g=spk_emb)
        x = self.linear(x).transpose(1, 2)  # [Batch, T_len, Out_dim]
        return x


class PhonemePredictor(nn.Module):
    """predictor.py:22-35"""

    def __init__(self, dict_size, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout):
        super().__init__()
        self.phoneme_predictor = RelativeEncoder(hidden_channels, filter_channels, n_heads, n_layers=n_layers,
                                                 kernel_size=kernel_size, p_dropout=p_dropout)
        self.ph_proj = HipConv1d(hidden_channels, dict_size, 1)

    def forward(self, x, x_mask):
        x = self.phoneme_predictor(x, x_mask)
        ph_pred = self.ph_proj(x)  # [Batch, Dict_size, T_len]
        return F.log_softmax(ph_pred, dim=1)
