"""HiFi-GAN period / scale discriminators on the MI355X-native kernels, with the reference's parameter
names and shapes (``modules/discriminator.py:13-75``: ``convs.{i}.weight_g / weight_v / bias`` and ``conv_post.*``), so a
reference checkpoint's ``mel_disc`` state loads unchanged.  Used by the training step only (BASELINE config 3).

* The parameter holders stay ``nn.Conv1d`` / ``nn.Conv2d`` under ``weight_norm`` (or ``spectral_norm``) exactly as in the
  reference; ``forward`` never calls them -- it fires their forward-pre hooks (which rebuild ``.weight`` from g / v,
  differentiably) and hands weight and bias to ``visinger_amd.autograd.disc_conv1d``.
* Dense convs (every period-discriminator layer, the first and the last two scale-discriminator layers) run on the MFMA conv
  engine; a stride-3 conv as the stride-1 conv of its three de-interleaved input phases (``StridedConv1dFn``).  The period
  discriminator's (k, 1) Conv2d over [B, C, H, p] is p independent 1-D convs: the waveform is folded once into [B*p, 1, H] and
  stays in that layout; feature maps are returned as [B, C, H, p] VIEWS of it (same values as the reference's tensors).
* The grouped stride-4 convs of the scale discriminator (4 input channels per group) run on the VALU kernels of
  ``csrc/grouped_conv.hip``, forward and both gradients.
* leaky_relu between the layers is a PyTorch-ROCm elementwise op (autograd records it).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import weight_norm, spectral_norm

from .. import _lib as L
from ..autograd import disc_conv1d
from .commons.utils import get_padding

LRELU_SLOPE = 0.1

# (out_channels, kernel, stride, groups, padding) of DiscriminatorS.convs -- discriminator.py:55-60
_SCALE_LAYERS = ((16, 15, 1, 1, 7), (64, 41, 4, 4, 20), (256, 41, 4, 16, 20), (1024, 41, 4, 64, 20), (1024, 41, 4, 256, 20),
                 (1024, 5, 1, 1, 2))
# out_channels of DiscriminatorP.convs -- discriminator.py:20-24 (the last one has stride 1)
_PERIOD_CHANNELS = (32, 128, 512, 1024, 1024)


def _live_params(conv, x):
    """(weight [C_out, C_in/groups, k], bias) of a weight- / spectral-normed conv holder: its forward-pre hooks recompute
    ``conv.weight`` from the underlying parameters, exactly what calling the module would do first"""
    w = conv.__dict__.get("_w_eff")       # folded for the whole network at the top of the pass (weight_bank.WeightBank.refresh)
    if w is None:
        for hook in conv._forward_pre_hooks.values():
            hook(conv, (x,))
        w = conv.weight
    return (w.squeeze(-1) if w.dim() == 4 else w), conv.bias


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
        L.require_gpu()
        b, c, t = x.shape
        p = self.period
        rem = t % p
        if rem:
            x = F.pad(x, (0, p - rem), "reflect")
        h = x.shape[2] // p
        # [B, 1, H, p] -> p independent columns: [B*p, 1, H]
        cur = x.view(b, c, h, p).permute(0, 3, 1, 2).reshape(b * p, c, h)

        def as_reference(y):          # [B*p, C, H'] -> the reference's [B, C, H', p] (a view)
            return y.view(b, p, y.shape[1], y.shape[2]).permute(0, 2, 3, 1)

        fmap = []
        for conv in self.convs:
            w, bias = _live_params(conv, cur)
            cur = F.leaky_relu(disc_conv1d(conv, cur, w, bias, conv.stride[0], conv.padding[0]), LRELU_SLOPE)
            fmap.append(as_reference(cur))
        w, bias = _live_params(self.conv_post, cur)
        cur = disc_conv1d(self.conv_post, cur, w, bias, 1, self.conv_post.padding[0])
        out = as_reference(cur)
        fmap.append(out)
        return torch.flatten(out, 1, -1), fmap


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
        L.require_gpu()
        fmap = []
        for conv in self.convs:
            w, bias = _live_params(conv, x)
            x = F.leaky_relu(disc_conv1d(conv, x, w, bias, conv.stride[0], conv.padding[0], conv.groups), LRELU_SLOPE)
            fmap.append(x)
        w, bias = _live_params(self.conv_post, x)
        x = disc_conv1d(self.conv_post, x, w, bias, 1, self.conv_post.padding[0])
        fmap.append(x)
        return torch.flatten(x, 1, -1), fmap
