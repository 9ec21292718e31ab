"""HiFi-GAN period / scale discriminators with the reference's parameter names and shapes
(``modules/discriminator.py:13-75``): ``convs.{i}.weight_g / weight_v / bias`` and ``conv_post.*``.

Used by the training step only (BASELINE config 3).  As SURVEY.md row a13 allows, this first version runs on plain
PyTorch-ROCm ops (strided (k,1) Conv2d, grouped stride-4 Conv1d): the conv engine in csrc/ is stride-1/ungrouped.
They are *not* part of the synthesis hot path and are not in the bench."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import weight_norm, spectral_norm

from .commons.utils import get_padding

LRELU_SLOPE = 0.1

# (out_channels, kernel, stride, groups, padding) of DiscriminatorS.convs -- discriminator.py:55-60
_SCALE_LAYERS = ((16, 15, 1, 1, 7), (64, 41, 4, 4, 20), (256, 41, 4, 16, 20), (1024, 41, 4, 64, 20), (1024, 41, 4, 256, 20),
                 (1024, 5, 1, 1, 2))
# out_channels of DiscriminatorP.convs -- discriminator.py:20-24 (the last one has stride 1)
_PERIOD_CHANNELS = (32, 128, 512, 1024, 1024)


class DiscriminatorP(nn.Module):
    """discriminator.py:13-47: fold the waveform into [B, 1, T/p, p] and run (k,1) convs along the first axis."""

    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        self.period = period
        self.use_spectral_norm = use_spectral_norm
        norm_f = weight_norm if use_spectral_norm == False else spectral_norm  # noqa: E712 (mirrors the reference flag)
        pad = (get_padding(kernel_size, 1), 0)
        layers, c_in = [], 1
        for i, c_out in enumerate(_PERIOD_CHANNELS):
            st = stride if i < len(_PERIOD_CHANNELS) - 1 else 1
            layers.append(norm_f(nn.Conv2d(c_in, c_out, (kernel_size, 1), (st, 1), padding=pad)))
            c_in = c_out
        self.convs = nn.ModuleList(layers)
        self.conv_post = norm_f(nn.Conv2d(c_in, 1, (3, 1), 1, padding=(1, 0)))

    def forward(self, x):
        b, c, t = x.shape
        rem = t % self.period
        if rem:
            x = F.pad(x, (0, self.period - rem), "reflect")
        x = x.view(b, c, -1, self.period)
        fmap = []
        for conv in self.convs:
            x = F.leaky_relu(conv(x), LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap


class DiscriminatorS(nn.Module):
    """discriminator.py:50-75"""

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        norm_f = weight_norm if use_spectral_norm == False else spectral_norm  # noqa: E712
        layers, c_in = [], 1
        for c_out, k, st, groups, pad in _SCALE_LAYERS:
            layers.append(norm_f(nn.Conv1d(c_in, c_out, k, st, groups=groups, padding=pad)))
            c_in = c_out
        self.convs = nn.ModuleList(layers)
        self.conv_post = norm_f(nn.Conv1d(c_in, 1, 3, 1, padding=1))

    def forward(self, x):
        fmap = []
        for conv in self.convs:
            x = F.leaky_relu(conv(x), LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap
