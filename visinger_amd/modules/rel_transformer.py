"""MI355X-native mirror of the reference's modules/rel_transformer.py:24-345 -- LayerNorm,
SinusoidalPositionalEmbedding, MultiHeadAttention, RelativeEncoder, FFN (same names, signatures, parameters).
(ConvReluNorm / RelativeTransformerEncoder/Decoder at rel_transformer.py:348-453 are not used by VISinger.)

Per encoder layer: q/k/v 1x1 convs write one [B, 3C, T] buffer (mask applied while staging x) -> streaming-softmax
attention with the banded relative terms (vs_relattn_fwd) -> conv_o -> LayerNorm kernel with the residual add
fused -> FFN conv k (mask in, ReLU out) -> 1x1 conv (mask in) -> LayerNorm kernel with the residual add and the
NEXT layer's `(x + g) * mask` fused.
"""
import math

import os

import torch
from torch import nn

from .. import _lib as L
from .. import autograd
from ..ops import ConvOp, layernorm_c, rel_attention, _off
from .hipconv import HipConv1d, mask2d, _forward_only_guard, drop_process_local_state


class LayerNorm(nn.Module):
    """rel_transformer.py:24-42: normalises dim 1 of [B, C, T]; biased variance, eps inside rsqrt."""

    def __init__(self, channels, eps=1e-4):
        super().__init__()
        self.channels = channels
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        if autograd.training_path(self):
            return autograd.layer_norm(self, x)
        _forward_only_guard(self)
        shp = x.shape
        x3 = x.contiguous().float().reshape(shp[0], shp[1], -1)
        return layernorm_c(x3, self.gamma, self.beta, eps=self.eps).reshape(shp)


class SinusoidalPositionalEmbedding(nn.Module):
    """rel_transformer.py:45-100 -- the fairseq-style table: row p holds sin(p * f_i) for the first half of the columns and cos(p * f_i) for the
    second, f_i = 10000^(-i / (half - 1)); an odd width gets a zero column, the padding row is zero.  Positions are integer bookkeeping
    (running count of the non-padding entries), bit-exact on the device (csrc/index_ops.hip); the table is evaluated in fp32 with the
    reference's operation order, so the looked-up rows equal the reference's bit for bit."""

    def __init__(self, embedding_dim, padding_idx, init_size=1024):
        super().__init__()
        self.embedding_dim, self.padding_idx = embedding_dim, padding_idx
        self.weights = self.get_embedding(init_size, embedding_dim, padding_idx)
        self.register_buffer('_float_tensor', torch.FloatTensor(1))          # (the reference's device / dtype anchor; part of its state_dict)

    @staticmethod
    def get_embedding(num_embeddings, embedding_dim, padding_idx=None):
        half = embedding_dim // 2
        inv_freq = (torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).exp()
        angle = torch.outer(torch.arange(num_embeddings, dtype=torch.float32), inv_freq)              # [positions, half]
        table = torch.zeros(num_embeddings, embedding_dim)
        table[:, :half] = angle.sin()
        table[:, half:2 * half] = angle.cos()
        if padding_idx is not None:
            table[padding_idx].zero_()
        return table

    def make_positions(tensor, padding_idx):
        from ..ops import make_positions as _make_positions_hip
        return _make_positions_hip(tensor, padding_idx)

    def forward(self, bsz, seq_len, input):
        rows_needed = self.padding_idx + 1 + seq_len
        if self.weights is None or self.weights.shape[0] < rows_needed:         # grow the table on demand
            self.weights = self.get_embedding(rows_needed, self.embedding_dim, self.padding_idx)
        self.weights = self.weights.to(self._float_tensor)
        positions = SinusoidalPositionalEmbedding.make_positions(input, self.padding_idx)
        # ([B * T, D] rows regrouped as [bsz, seq_len, -1]: the text encoder calls this with seq_len = D, see TextEncoder.forward_text_embedding)
        return torch.nn.functional.embedding(positions.reshape(-1), self.weights).view(bsz, seq_len, -1).detach()


class MultiHeadAttention(nn.Module):
    """rel_transformer.py:103-254"""

    def __init__(self, channels, out_channels, n_heads, window_size=None, heads_share=True, p_dropout=0.,
                 block_length=None, proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        self.channels = channels
        self.out_channels = out_channels
        self.n_heads = n_heads
        self.window_size = window_size
        self.heads_share = heads_share
        self.block_length = block_length
        self.proximal_bias = proximal_bias
        self.p_dropout = p_dropout
        # `attn` -- the reference's non-persistent module state, the [B, h, T, T] probabilities of the last forward (rel_transformer.py:143, 171) -- is not
        # materialised by the streaming kernels (O(T^2) memory that nothing on the path reads).  `store_attn = True` (per module) routes the forward through the
        # [T, T] core on PyTorch-ROCm ops and keeps them, as the reference does on every call; pinned to the reference's own `attn` (tests/test_modules_gpu.py).
        self.attn = None
        self.store_attn = False

        self.k_channels = channels // n_heads
        self.conv_q = HipConv1d(channels, channels, 1)
        self.conv_k = HipConv1d(channels, channels, 1)
        self.conv_v = HipConv1d(channels, channels, 1)
        if window_size is not None:
            n_heads_rel = 1 if heads_share else n_heads
            rel_stddev = self.k_channels ** -0.5
            self.emb_rel_k = nn.Parameter(torch.randn(n_heads_rel, window_size * 2 + 1, self.k_channels) * rel_stddev)
            self.emb_rel_v = nn.Parameter(torch.randn(n_heads_rel, window_size * 2 + 1, self.k_channels) * rel_stddev)
        self.conv_o = HipConv1d(channels, out_channels, 1)
        self.drop = nn.Dropout(p_dropout)

        nn.init.xavier_uniform_(self.conv_q.weight)
        nn.init.xavier_uniform_(self.conv_k.weight)
        if proximal_init:
            self.conv_k.weight.data.copy_(self.conv_q.weight.data)
            self.conv_k.bias.data.copy_(self.conv_q.bias.data)
        nn.init.xavier_uniform_(self.conv_v.weight)

    def __getstate__(self):      # the fused q | k | v handle and its concatenated device copies are process-local caches
        return drop_process_local_state(self.__dict__.copy())

    def _fused_qkv_op(self):
        """One conv handle for conv_q | conv_k | conv_v (three nn.Conv1d(channels, channels, 1) of the same input in self-attention,
        rel_transformer.py:120-122, 141-143): the weights are concatenated and packed once per version of the six parameters."""
        convs = (self.conv_q, self.conv_k, self.conv_v)
        if any(hasattr(cv, "weight_g") or cv.bias is None for cv in convs):
            return None
        params = [t for cv in convs for t in (cv.weight, cv.bias)]
        math = self.conv_q._op(bind=False).math
        key = tuple((t.data_ptr(), t._version) for t in params) + (math,)
        st = self.__dict__.get("_hip_qkv_inf")
        if st is None or st[0] != key:
            op = st[1] if st is not None else ConvOp(L.CONV1D, self.channels, 3 * self.channels, 1, 1, 0)
            if op.math != math:
                op.set_math(math)
            w = torch.cat([cv.weight.detach() for cv in convs], 0).contiguous()
            b = torch.cat([cv.bias.detach() for cv in convs], 0).contiguous()
            op.set_weights(w, None, b, force=True)
            st = self.__dict__["_hip_qkv_inf"] = (key, op, w, b)
        return st[1]

    def forward(self, x, c, attn_mask=None, frame_mask=None, in_mask=False):
        """x, c: [B, C, T].  attn_mask: the reference's [B, 1, T, T] mask, which RelativeEncoder always builds as
        m[:, :, :, None] * m[:, :, None, :] from the frame mask m; the streaming kernel takes m itself (`frame_mask`
        [B, T]); a 4-D attn_mask is reduced back to m through its diagonal.  in_mask: multiply x by m while
        staging (fuses RelativeEncoder's `x = x * x_mask`)."""
        assert x.shape == c.shape, "Relative attention is only available for self-attention."
        B, C, T = x.shape
        if frame_mask is None and attn_mask is not None:
            frame_mask = torch.diagonal(attn_mask.reshape(B, T, T), dim1=1, dim2=2)
        if autograd.training_path(self):
            assert c is x, "the training path implements self-attention"
            return autograd.attention(self, x * frame_mask.reshape(B, 1, T) if (in_mask and frame_mask is not None) else x,
                                      None if frame_mask is None else frame_mask.reshape(B, T).float())
        _forward_only_guard(self)
        if self.k_channels > 256 or self.proximal_bias or self.block_length is not None or self.__dict__.get("store_attn", False):
            # (proximal_bias / block_length -- rel_transformer.py:163-170, never set by VISinger -- are not in the streaming kernels: the q / k / v / o convs on
            #  the HIP engine, the [T, T] core with the two options as PyTorch-ROCm ops, like the heads wider than 256 channels)
            # the streaming kernel covers heads of up to 256 channels (BASELINE config 5: hidden 512, 2 heads; the query tile
            # moves to LDS above 128); anything wider runs the q/k/v/o convs on the HIP engine and the [T, T] core as
            # PyTorch-ROCm ops (same index arithmetic, same -1e4 mask fill).
            assert c is x, "wide-head path implements self-attention"
            with torch.no_grad():
                fm = None if frame_mask is None else frame_mask.reshape(B, T).float()
                return autograd.attention(self, x * fm.reshape(B, 1, T) if (in_mask and fm is not None) else x, fm)
        m2 = None if frame_mask is None else frame_mask.reshape(B, T).float().contiguous()
        x = x.contiguous().float()
        c = x if c is x else c.contiguous().float()
        qkv = torch.empty((B, 3 * C, T), device=x.device, dtype=torch.float32)
        ia = L.IN_MASK if in_mask else L.IN_NONE
        fused = self._fused_qkv_op() if (c is x and not L.switch("VS_NO_FUSED_QKV")) else None
        if fused is not None:
            fused.forward(x, in_act=ia, mask=m2, y=qkv)       # q | k | v: ONE [3C, C] projection of the same x (rows are independent: same sums)
        else:
            for j, (conv, src) in enumerate(((self.conv_q, x), (self.conv_k, c), (self.conv_v, c))):
                conv.run(src, in_act=ia, mask=m2, y_ptr=_off(qkv, j * C * T), y_bs=3 * C * T)
        rel_k = self.emb_rel_k if self.window_size is not None else None
        rel_v = self.emb_rel_v if self.window_size is not None else None
        # the attention core follows the arithmetic of the projections around it (bf16 GEMMs only under VS_MATH_BF16)
        att = rel_attention(qkv, self.n_heads, rel_k, rel_v, m2, self.window_size, math=self.conv_q._op(bind=False).math)
        return self.conv_o.run(att)


class RelativeEncoder(nn.Module):
    """rel_transformer.py:257-320"""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.,
                 window_size=4, block_length=None, pre_ln=False, gin_channels=None, **kwargs):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.filter_channels = filter_channels
        self.n_heads = n_heads
        self.n_layers = n_layers
        self.kernel_size = kernel_size
        self.p_dropout = p_dropout
        self.window_size = window_size
        self.block_length = block_length
        self.pre_ln = pre_ln

        self.drop = nn.Dropout(p_dropout)
        self.attn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_2 = nn.ModuleList()
        for i in range(self.n_layers):
            self.attn_layers.append(
                MultiHeadAttention(hidden_channels, hidden_channels, n_heads, window_size=window_size,
                                   p_dropout=p_dropout, block_length=block_length))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(
                FFN(hidden_channels, hidden_channels, filter_channels, kernel_size, p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))
        if pre_ln:
            self.last_ln = LayerNorm(hidden_channels)
        if gin_channels is not None:
            self.pre_net = HipConv1d(gin_channels, hidden_channels, 1)

    def forward(self, x, x_mask, g=None):
        if autograd.training_path(self):
            return autograd.rel_encoder(self, x, x_mask.reshape(x.shape[0], 1, x.shape[2]), g)
        _forward_only_guard(self)
        if self.pre_ln:     # (rel_transformer.py:301-317, never set by VISinger: the same composition as the training path -- HIP convs / LayerNorm / attention
            with torch.no_grad():       # kernels, the residual adds as PyTorch-ROCm ops -- instead of the fused post-LN launch sequence below)
                return autograd.rel_encoder(self, x.contiguous().float(), x_mask.reshape(x.shape[0], 1, x.shape[2]).float(),
                                            None if g is None else g.contiguous().float())
        B, C, T = x.shape
        m2 = mask2d(x_mask, B, T)
        x = x.contiguous().float()
        if g is not None:
            g = self.pre_net.run(g.contiguous().float())          # [B, C, T] or [B, C, 1]
            x = x + g                                             # first layer's conditioning add (:297-298)
        x = x * m2[:, None, :]                                    # (:299)
        for i in range(self.n_layers):
            # attention block, post-LN: x = LN1(x + attn(x))
            y = self.attn_layers[i](x, x, frame_mask=m2)
            ln1 = self.norm_layers_1[i]
            x = layernorm_c(y, ln1.gamma, ln1.beta, r=x, eps=ln1.eps)
            # FFN block, post-LN: x = LN2(x + ffn(x)); then the next layer's (x + g) * mask (or the final mask)
            y = self.ffn_layers[i](x, m2)
            ln2 = self.norm_layers_2[i]
            last = (i == self.n_layers - 1)
            x = layernorm_c(y, ln2.gamma, ln2.beta, r=x, g=None if last else g, mask=m2, eps=ln2.eps)
        return x


class FFN(nn.Module):
    """rel_transformer.py:323-345"""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0, activation=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.filter_channels = filter_channels
        self.kernel_size = kernel_size
        self.activation = activation
        self.conv_1 = HipConv1d(in_channels, filter_channels, kernel_size, padding=kernel_size // 2)
        self.conv_2 = HipConv1d(filter_channels, out_channels, 1)
        self.dropout = nn.Dropout(p_dropout)

    def forward(self, x, x_mask):
        if autograd.training_path(self):
            return autograd.ffn(self, x, x_mask.reshape(x.shape[0], 1, x.shape[2]))
        _forward_only_guard(self)
        B, _, T = x.shape
        m2 = mask2d(x_mask, B, T)
        if self.activation == "gelu":       # (rel_transformer.py:338-341, never built by VISinger: x * sigmoid(1.702 x) between the two HIP convs as PyTorch-ROCm ops)
            h = self.conv_1.run(x.contiguous().float(), in_act=L.IN_MASK, mask=m2)
            h = h * torch.sigmoid(1.702 * h)
            return self.conv_2.run(h, in_act=L.IN_MASK, mask=m2)
        h = self.conv_1.run(x.contiguous().float(), in_act=L.IN_MASK, mask=m2, out_act=L.OUT_RELU)
        return self.conv_2.run(h, in_act=L.IN_MASK, mask=m2)
