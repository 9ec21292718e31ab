"""MI355X-native mirror of the reference's modules/visinger/encoder.py (same class names, constructor and
forward signatures, parameter names) -- WaveNet, PosteriorEncoder, FramePriorNetwork, TextEncoder.

Arithmetic runs in libvisinger_hip.so (gfx950); this file is host-side plumbing."""
import math

import torch
import torch.nn as nn

from ... import _lib as L
from ... import autograd
from ...models.commons.align_ops import expand_states  # noqa: F401  (re-exported like the reference module)
from ...ops import expand_states as _expand_states_hip
from ..commons.utils import Embedding
from ..hipconv import HipConv1d, mask2d, _forward_only_guard
from ..rel_transformer import RelativeEncoder, SinusoidalPositionalEmbedding

DEFAULT_MAX_TARGET_POSITIONS = 2000
LRELU_SLOPE = 0.1


class WaveNet(torch.nn.Module):
    """encoder.py:130-203.  Per layer two launches: the dilated k-tap conv with the conditioning add and the
    tanh*sigmoid gate fused into its epilogue, and the 1x1 res/skip conv whose epilogue writes both destinations
    (x = (x + res) * mask ; output += skip)."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super(WaveNet, self).__init__()
        assert (kernel_size % 2 == 1)
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size,
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.gin_channels = gin_channels
        self.p_dropout = p_dropout

        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        self.drop = nn.Dropout(p_dropout)

        if gin_channels != 0:
            cond_layer = HipConv1d(gin_channels, 2 * hidden_channels * n_layers, 1)
            self.cond_layer = nn.utils.weight_norm(cond_layer, name='weight')

        for i in range(n_layers):
            dilation = dilation_rate ** i
            padding = int((kernel_size * dilation - dilation) / 2)
            in_layer = HipConv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=dilation, padding=padding)
            in_layer = nn.utils.weight_norm(in_layer, name='weight')
            self.in_layers.append(in_layer)
            res_skip_channels = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            res_skip_layer = HipConv1d(hidden_channels, res_skip_channels, 1)
            res_skip_layer = nn.utils.weight_norm(res_skip_layer, name='weight')
            self.res_skip_layers.append(res_skip_layer)

    def forward(self, x, x_mask, g=None, **kwargs):
        if autograd.training_path(self):
            return autograd.wavenet(self, x, x_mask.reshape(x.shape[0], 1, x.shape[2]), g)
        _forward_only_guard(self)
        B, H, T = x.shape
        x = x.contiguous().float()
        m2 = mask2d(x_mask, B, T)
        L_ = self.n_layers
        gc = None
        if g is not None:
            gc = self.cond_layer.run(g.contiguous().float())          # [B, 2H*L, 1]
        xbuf = None
        out = torch.empty_like(x)
        acts = torch.empty_like(x)
        cur = x
        for i in range(L_):
            kw = {}
            if gc is not None:
                # this layer's conditioning slice g[:, i*2H:(i+1)*2H] (encoder.py:177-178): a pointer i*2H floats
                # into each item's row of the cond_layer output, row stride 2H*L
                kw = dict(bias_b=gc.view(-1)[i * 2 * H:], bias_b_bs=2 * H * L_)
            self.in_layers[i].run(cur, kind=L.CONV1D_PAIRED, pair_mode=L.PAIR_GATE, y=acts, **kw)
            if i < L_ - 1:
                if xbuf is None:
                    xbuf = torch.empty_like(x)
                self.res_skip_layers[i].run(acts, y=xbuf, res=cur, out_mask=True, mask=m2, split_row=H,
                                            out1=dict(y=out, acc=out if i > 0 else None))
                cur = xbuf
            else:
                self.res_skip_layers[i].run(acts, y=out, acc=out if i > 0 else None, out_mask=True, mask=m2)
        return out

    def remove_weight_norm(self):
        if self.gin_channels != 0:
            torch.nn.utils.remove_weight_norm(self.cond_layer)
        for l in self.in_layers:
            torch.nn.utils.remove_weight_norm(l)
        for l in self.res_skip_layers:
            torch.nn.utils.remove_weight_norm(l)


class PosteriorEncoder(nn.Module):
    """encoder.py:76-101"""

    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.dilation_rate = dilation_rate
        self.n_layers = n_layers
        self.gin_channels = gin_channels

        self.pre = HipConv1d(in_channels, hidden_channels, 1)
        self.enc = WaveNet(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = HipConv1d(hidden_channels, out_channels * 2, 1)

    def forward(self, x, nonpadding, g=None, noise=None):
        """`noise` (optional, [B, out_channels, T]) replaces torch.randn_like(mu_q) for reproducible parity."""
        if autograd.training_path(self):
            h = autograd.conv(self.pre, x) * nonpadding
            h = self.enc(h, nonpadding, g=g)
            stats = autograd.conv(self.proj, h) * nonpadding
            mu_q, logs_q = torch.split(stats, self.out_channels, dim=1)
            noise = torch.randn_like(mu_q) if noise is None else noise
            return (mu_q + noise * torch.exp(logs_q)) * nonpadding, mu_q, logs_q
        B, _, T = x.shape
        m2 = mask2d(nonpadding, B, T)
        h = self.pre.run(x.contiguous().float(), mask=m2, out_mask=True)
        h = self.enc(h, nonpadding, g=g)
        stats = self.proj.run(h, mask=m2, out_mask=True)
        mu_q, logs_q = torch.split(stats, self.out_channels, dim=1)
        if noise is None:
            noise = torch.randn_like(mu_q)
        z_q = (mu_q + noise * torch.exp(logs_q)) * nonpadding
        return z_q, mu_q, logs_q

    def remove_weight_norm(self):
        self.enc.remove_weight_norm()


class FramePriorNetwork(nn.Module):
    """encoder.py:58-73"""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, gin_channels, p_dropout):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.encoder = RelativeEncoder(hidden_channels, filter_channels, n_heads, n_layers=n_layers,
                                       kernel_size=kernel_size, gin_channels=gin_channels, p_dropout=p_dropout)
        self.proj = HipConv1d(self.hidden_channels, self.hidden_channels * 2, 1)

    def forward(self, x, x_mask, g=None):
        if g is not None:
            g = g.transpose(1, 2)           # as the reference does (encoder.py:68-69)
        prior_out = self.encoder(x, x_mask, g)
        B, _, T = prior_out.shape
        if autograd.training_path(self):
            prior_out = autograd.conv(self.proj, prior_out) * x_mask
        else:
            prior_out = self.proj.run(prior_out, mask=mask2d(x_mask, B, T), out_mask=True)
        mu_p, logs_p = torch.split(prior_out, self.hidden_channels, dim=1)
        return mu_p, logs_p


class TextEncoder(nn.Module):
    """encoder.py:14-55"""

    def __init__(self, ph_dict_size, note_pitch_size, note_dur_size, hidden_channels, filter_channels,
                 n_heads, n_layers, kernel_size, p_dropout, use_pos_embed=False):
        super().__init__()
        self.dropout = p_dropout
        self.use_pos_embed = use_pos_embed
        self.ph_emb = Embedding(ph_dict_size, hidden_channels)
        self.pitch_emb = Embedding(note_pitch_size, hidden_channels)
        self.dur_emb = Embedding(note_dur_size, hidden_channels)
        self.embed_scale = math.sqrt(hidden_channels)
        self.padding_idx = 0
        self.linear = nn.Linear(hidden_channels * 3, hidden_channels)
        self.text_encoder = RelativeEncoder(hidden_channels, filter_channels, n_heads, n_layers,
                                            kernel_size=kernel_size, p_dropout=p_dropout)
        if self.use_pos_embed:
            self.embed_positions = SinusoidalPositionalEmbedding(hidden_channels, 0, init_size=DEFAULT_MAX_TARGET_POSITIONS)

    def forward(self, text_tokens, pitch_tokens, dur_tokens, mel2ph):
        tgt_nonpadding = (text_tokens > 0).float().unsqueeze(1)
        token_emb = self.forward_text_embedding(text_tokens, pitch_tokens, dur_tokens, tgt_nonpadding.transpose(1, 2))
        enc_out = self.text_encoder(token_emb.transpose(1, 2), tgt_nonpadding)       # [B, H, T_ph]
        if autograd.training_path(self):      # differentiable gather (PyTorch-ROCm), same indexing
            hp = torch.nn.functional.pad(enc_out, [1, 0])
            return torch.gather(hp, 2, mel2ph[:, None, :].expand(-1, hp.shape[1], -1))
        # expand_states(enc_out.transpose(1, 2), mel2ph).transpose(1, 2) without the two transposes (encoder.py:39-40)
        return _expand_states_hip(enc_out, mel2ph, h_channels_first=True, out_channels_first=True)

    def forward_text_embedding(self, text_tokens, pitch_tokens, dur_tokens, nonpadding):
        """encoder.py:42-55: the three token streams embedded, scaled by sqrt(H), concatenated and mixed by one Linear; with use_pos_embed the
        sinusoidal rows of the running token count are added.  T_ph-sized glue on plain PyTorch-ROCm ops (negligible next to the encoder)."""
        streams = ((self.ph_emb, text_tokens), (self.pitch_emb, pitch_tokens), (self.dur_emb, dur_tokens))
        mixed = self.linear(torch.cat([table(ids) * self.embed_scale for table, ids in streams], dim=2)) * nonpadding      # [B, T_ph, H]
        if self.use_pos_embed:
            # the reference hands embed_positions seq_len = mixed.shape[2] (the hidden width, not T_ph) and transposes the result: reproduced
            # literally -- it type-checks because B * T_ph * H elements come back either way (encoder.py:52-54)
            batch, _, width = mixed.shape
            mixed = mixed + self.embed_positions(batch, width, mixed[..., 0]).transpose(1, 2)
        return mixed * nonpadding
