"""CPU oracle for VISinger's variational-inference hot path (numpy composition over the C primitives in
vs_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of vs_oracle.c.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py import this module; the product package visinger_amd never does.

Parity: PINNED against tests/golden/*.npz (vectors produced by importing the reference, see
tests/golden/make_golden.py) by tests/test_oracle_golden.py.

Every function takes the module's weights as a dict keyed exactly like the reference's ``state_dict()``
(old-style weight-norm pairs ``*.weight_g`` / ``*.weight_v``), so a fixture or a checkpoint feeds it directly.
``dtype`` selects the arithmetic: np.float64 (referee) or np.float32 (the "port" CPU baseline).

Each function cites the reference lines it restates (paths under /root/reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
LRELU_SLOPE = 0.1  # modules/visinger/decoder.py:10


def build(force=False):
    """Compile vs_oracle.c -> oracle/_build/libvs_oracle_{f64,f32}.so (gcc, OpenMP)."""
    outs = [os.path.join(_HERE, "_build", f"libvs_oracle_{t}.so") for t in ("f64", "f32")]
    src = os.path.join(_HERE, "vs_oracle.c")
    if force or not all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return outs


def _lib(dtype):
    dtype = np.dtype(dtype)
    tag = {np.dtype(np.float64): "f64", np.dtype(np.float32): "f32"}[dtype]
    if tag not in _LIBS:
        path = os.path.join(_HERE, "_build", f"libvs_oracle_{tag}.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        assert lib.orc_real_bytes() == dtype.itemsize
        _LIBS[tag] = lib
    return _LIBS[tag]


def set_threads(n):
    """OpenMP threads of the C primitives (both precisions); returns the count in effect."""
    return min(_lib(np.float64).orc_set_threads(int(n)), _lib(np.float32).orc_set_threads(int(n)))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


_i64 = ctypes.c_int64

# "c" = the C restatement below (the referee, and the `port` CPU baseline); "torch" = the same composition with the two
# convolution primitives executed by stock PyTorch CPU kernels (torch.nn.functional, oneDNN) -- the second CPU baseline
# BASELINE.md 4 asks for ("stock-PyTorch CPU model of identical architecture"), used by bench.py only.
CONV_BACKEND = "c"


# Operand rounding (tests only): None = the reference's arithmetic; "bf16" = every matrix-product operand (conv inputs after their input
# transform, effective conv weights, q / k / v of the attention core) rounded to bfloat16, round-to-nearest-even, products and sums in
# `dtype` -- the arithmetic BASELINE configs[4] names ("bf16 activations/weights, fp32 accumulate") and the device's VS_MATH_BF16
# computes.  A parity test of that configuration compares against THIS, so that what is left is accumulation order and roundings that
# fall on a bf16 boundary, not the 2^-9 operand error itself (VERDICT r5 next #1c).
OPERAND_ROUNDING = None


class operand_rounding:
    """with operand_rounding("bf16"): ... -- scoped setting of OPERAND_ROUNDING"""

    def __init__(self, mode):
        assert mode in (None, "bf16")
        self.mode = mode

    def __enter__(self):
        global OPERAND_ROUNDING
        self.prev, OPERAND_ROUNDING = OPERAND_ROUNDING, self.mode

    def __exit__(self, *exc):
        global OPERAND_ROUNDING
        OPERAND_ROUNDING = self.prev


def round_bf16(a):
    """round-to-nearest-even to bfloat16, returned in a's dtype (NaN / inf pass through)"""
    a = np.asarray(a)
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    r = np.where(np.isfinite(f), r, f)
    return r.astype(a.dtype if a.dtype in (np.float32, np.float64) else np.float32)


def _operands(*arrays):
    if OPERAND_ROUNDING is None:
        return arrays
    return tuple(None if a is None else round_bf16(a) for a in arrays)


def _torch_conv(x, w, bias, transposed, **kw):
    import torch
    import torch.nn.functional as F
    t = [None if a is None else torch.from_numpy(a) for a in (x, w, bias)]
    with torch.no_grad():
        y = (F.conv_transpose1d if transposed else F.conv1d)(t[0], t[1], t[2], **kw)
    return y.numpy()


# ------------------------------------------------------------------------------------------------------------
# primitives


def weight_norm(v, g, dtype=np.float64):
    """w = g * v / ||v||_2, norm over every dim but 0 (torch.nn.utils.weight_norm, dim=0); used at
    encoder.py:147,154,164, decoder.py:24,72-87.  For ConvTranspose1d dim 0 is the INPUT channel."""
    v = _c(v, dtype)
    g = _c(np.reshape(g, -1), dtype)
    w = np.empty_like(v)
    rows = v.shape[0]
    _lib(dtype).orc_weightnorm(_p(v), _p(g), _p(w), _i64(rows), _i64(v.size // rows))
    return w


def _get_w(sd, name, dtype):
    """Effective weight of conv `name`: plain ``name.weight`` or folded ``weight_g/weight_v`` pair."""
    if name + ".weight" in sd:
        return _c(sd[name + ".weight"], dtype)
    return weight_norm(sd[name + ".weight_v"], sd[name + ".weight_g"], dtype)


def conv1d(x, w, bias=None, dilation=1, padding=0, dtype=np.float64):
    """nn.Conv1d, stride 1, zero padding (all Conv1d sites of the hot path)."""
    x, w, bias = _c(x, dtype), _c(w, dtype), _c(bias, dtype)
    x, w = _operands(x, w)
    B, Cin, T = x.shape
    Cout, Cin2, K = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    Tout = T + 2 * padding - dilation * (K - 1)
    if CONV_BACKEND == "torch":
        return _torch_conv(x, w, bias, False, dilation=dilation, padding=padding)
    y = np.empty((B, Cout, Tout), dtype=dtype)
    _lib(dtype).orc_conv1d(_p(x), _p(w), _p(bias), _p(y), _i64(B), _i64(Cin), _i64(Cout), _i64(T), _i64(K),
                           _i64(dilation), _i64(padding))
    return y


def conv_transpose1d(x, w, bias=None, stride=1, padding=0, dtype=np.float64):
    """nn.ConvTranspose1d (decoder.py:24-26,47); w is [Cin, Cout, K]."""
    x, w, bias = _c(x, dtype), _c(w, dtype), _c(bias, dtype)
    x, w = _operands(x, w)
    B, Cin, T = x.shape
    Cin2, Cout, K = w.shape
    assert Cin == Cin2
    Tout = (T - 1) * stride - 2 * padding + K
    if CONV_BACKEND == "torch":
        return _torch_conv(x, w, bias, True, stride=stride, padding=padding)
    y = np.empty((B, Cout, Tout), dtype=dtype)
    _lib(dtype).orc_conv_transpose1d(_p(x), _p(w), _p(bias), _p(y), _i64(B), _i64(Cin), _i64(Cout), _i64(T),
                                     _i64(K), _i64(stride), _i64(padding))
    return y


def conv1d_sg(x, w, bias=None, stride=1, padding=0, groups=1, dtype=np.float64):
    """Strided / grouped nn.Conv1d (discriminator.py:55-63); also serves the (k,1)-kernel Conv2d of DiscriminatorP
    (discriminator.py:20-26) applied column by column."""
    x, w, bias = _c(x, dtype), _c(w, dtype), _c(bias, dtype)
    B, Cin, T = x.shape
    Cout, cig, K = w.shape
    assert Cin == cig * groups
    Tout = (T + 2 * padding - K) // stride + 1
    y = np.empty((B, Cout, Tout), dtype=dtype)
    _lib(dtype).orc_conv1d_sg(_p(x), _p(w), _p(bias), _p(y), _i64(B), _i64(Cin), _i64(Cout), _i64(T), _i64(K), _i64(stride),
                              _i64(padding), _i64(groups))
    return y


def layer_norm_c(x, gamma, beta, eps=1e-4, dtype=np.float64):
    """rel_transformer.py:24-42 (channel LayerNorm, biased variance, eps=1e-4)."""
    x, gamma, beta = _c(x, dtype), _c(gamma, dtype), _c(beta, dtype)
    B, C, T = x.shape
    y = np.empty_like(x)
    _lib(dtype).orc_layernorm_c(_p(x), _p(gamma), _p(beta), _p(y), _i64(B), _i64(C), _i64(T), ctypes.c_double(eps))
    return y


def leaky_relu(x, slope=LRELU_SLOPE):
    return np.where(x >= 0, x, x * x.dtype.type(slope))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def get_padding(kernel_size, dilation=1):
    """modules/commons/utils.py:109-110"""
    return int((kernel_size * dilation - dilation) / 2)


def _sub(sd, prefix):
    """state-dict entries under `prefix.` with the prefix stripped."""
    if not prefix:
        return sd
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


# ------------------------------------------------------------------------------------------------------------
# WaveNet / posterior encoder / flow


def wavenet(sd, x, x_mask, g=None, *, hidden_channels, kernel_size, dilation_rate, n_layers, dtype=np.float64):
    """WaveNet.forward, encoder.py:167-195 (gate: encoder.py:206-213)."""
    H = hidden_channels
    x = _c(x, dtype)
    x_mask = _c(x_mask, dtype)
    output = np.zeros_like(x)
    if g is not None:
        g = conv1d(g, _get_w(sd, "cond_layer", dtype), sd["cond_layer.bias"], dtype=dtype)       # :172
    for i in range(n_layers):
        dil = dilation_rate ** i
        pad = int((kernel_size * dil - dil) / 2)                                                  # :150-151
        x_in = conv1d(x, _get_w(sd, f"in_layers.{i}", dtype), sd[f"in_layers.{i}.bias"], dilation=dil, padding=pad,
                      dtype=dtype)                                                                 # :175
        if g is not None:
            x_in = x_in + g[:, i * 2 * H:(i + 1) * 2 * H, :]                                      # :177-178
        acts = np.tanh(x_in[:, :H]) * sigmoid(x_in[:, H:])                                        # :206-213
        rs = conv1d(acts, _get_w(sd, f"res_skip_layers.{i}", dtype), sd[f"res_skip_layers.{i}.bias"], dtype=dtype)
        if i < n_layers - 1:
            x = (x + rs[:, :H]) * x_mask                                                           # :190-191
            output = output + rs[:, H:]                                                            # :192
        else:
            output = output + rs                                                                   # :194
    return output * x_mask


def posterior_encoder(sd, x, nonpadding, g, noise, *, out_channels, hidden_channels, kernel_size, dilation_rate,
                      n_layers, dtype=np.float64):
    """PosteriorEncoder.forward, encoder.py:92-98; `noise` replaces torch.randn_like(mu_q)."""
    nonpadding = _c(nonpadding, dtype)
    h = conv1d(x, sd["pre.weight"], sd["pre.bias"], dtype=dtype) * nonpadding
    h = wavenet(_sub(sd, "enc"), h, nonpadding, g, hidden_channels=hidden_channels, kernel_size=kernel_size,
                dilation_rate=dilation_rate, n_layers=n_layers, dtype=dtype)
    stats = conv1d(h, sd["proj.weight"], sd["proj.bias"], dtype=dtype) * nonpadding
    mu, logs = stats[:, :out_channels], stats[:, out_channels:]
    z = (mu + _c(noise, dtype) * np.exp(logs)) * nonpadding
    return z, mu, logs


def coupling_layer(sd, x, x_mask, g=None, reverse=False, *, channels, hidden_channels, kernel_size, dilation_rate,
                   n_layers, mean_only=False, dtype=np.float64):
    """ResidualCouplingLayer.forward, flow.py:66-85.  Returns (x, logdet) forward, x in reverse."""
    half = channels // 2
    x = _c(x, dtype)
    x_mask = _c(x_mask, dtype)
    x0, x1 = x[:, :half], x[:, half:]
    h = conv1d(x0, sd["pre.weight"], sd["pre.bias"], dtype=dtype) * x_mask
    h = wavenet(_sub(sd, "enc"), h, x_mask, g, hidden_channels=hidden_channels, kernel_size=kernel_size,
                dilation_rate=dilation_rate, n_layers=n_layers, dtype=dtype)
    stats = conv1d(h, sd["post.weight"], sd["post.bias"], dtype=dtype) * x_mask
    if not mean_only:
        m, logs = stats[:, :half], stats[:, half:]
    else:
        m, logs = stats, np.zeros_like(stats)
    if not reverse:
        x1 = m + x1 * np.exp(logs) * x_mask
        return np.concatenate([x0, x1], 1), logs.sum(axis=(1, 2))
    x1 = (x1 - m) * np.exp(-logs) * x_mask
    return np.concatenate([x0, x1], 1)


def flow_block(sd, x, x_mask, g=None, reverse=False, *, channels, hidden_channels, kernel_size, dilation_rate,
               n_layers, n_flows=4, mean_only=True, return_logdet=False, dtype=np.float64):
    """ResidualCouplingBlock.forward, flow.py:33-40 (couplings at flows.{0,2,..}, Flip at the odd slots).
    The reference drops the per-layer log-det (`x, _ = flow(...)`); return_logdet=True also returns their sum."""
    kw = dict(channels=channels, hidden_channels=hidden_channels, kernel_size=kernel_size, dilation_rate=dilation_rate,
              n_layers=n_layers, mean_only=mean_only, dtype=dtype)
    x = _c(x, dtype)
    logdet = np.zeros(x.shape[0], dtype=dtype)
    if not reverse:
        for f in range(n_flows):
            x, ld = coupling_layer(_sub(sd, f"flows.{2 * f}"), x, x_mask, g, False, **kw)
            logdet = logdet + ld
            x = x[:, ::-1]                                   # Flip, flow.py:90
    else:
        for f in reversed(range(n_flows)):
            x = x[:, ::-1]
            x = coupling_layer(_sub(sd, f"flows.{2 * f}"), x, x_mask, g, True, **kw)
    x = np.ascontiguousarray(x)
    return (x, logdet) if return_logdet else x


# ------------------------------------------------------------------------------------------------------------
# HiFi-GAN generator


def resblock1(sd, x, x_mask=None, *, kernel_size, dilation=(1, 3, 5), dtype=np.float64):
    """ResBlock1.forward, decoder.py:91-104."""
    x = _c(x, dtype)
    for i, d in enumerate(dilation):
        xt = leaky_relu(x)
        if x_mask is not None:
            xt = xt * x_mask
        xt = conv1d(xt, _get_w(sd, f"convs1.{i}", dtype), sd[f"convs1.{i}.bias"], dilation=d,
                    padding=get_padding(kernel_size, d), dtype=dtype)
        xt = leaky_relu(xt)
        if x_mask is not None:
            xt = xt * x_mask
        xt = conv1d(xt, _get_w(sd, f"convs2.{i}", dtype), sd[f"convs2.{i}.bias"], dilation=1,
                    padding=get_padding(kernel_size, 1), dtype=dtype)
        x = xt + x
    if x_mask is not None:
        x = x * x_mask
    return x


def resblock2(sd, x, x_mask=None, *, kernel_size, dilation=(1, 3), dtype=np.float64):
    """ResBlock2.forward, decoder.py:124-133."""
    x = _c(x, dtype)
    for i, d in enumerate(dilation):
        xt = leaky_relu(x)
        if x_mask is not None:
            xt = xt * x_mask
        xt = conv1d(xt, _get_w(sd, f"convs.{i}", dtype), sd[f"convs.{i}.bias"], dilation=d,
                    padding=get_padding(kernel_size, d), dtype=dtype)
        x = xt + x
    if x_mask is not None:
        x = x * x_mask
    return x


def generator(sd, x, g=None, *, resblock="1", resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
              upsample_kernel_sizes, dtype=np.float64):
    """Generator.forward, decoder.py:40-59."""
    nk = len(resblock_kernel_sizes)
    x = conv1d(x, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3, dtype=dtype)
    if g is not None:
        x = x + conv1d(g, sd["cond.weight"], sd["cond.bias"], dtype=dtype)
    rb = resblock1 if str(resblock) == "1" else resblock2
    for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
        x = leaky_relu(x)
        x = conv_transpose1d(x, _get_w(sd, f"ups.{i}", dtype), sd[f"ups.{i}.bias"], stride=int(u),
                             padding=(int(k) - int(u)) // 2, dtype=dtype)
        xs = None
        for j in range(nk):
            r = rb(_sub(sd, f"resblocks.{i * nk + j}"), x, kernel_size=int(resblock_kernel_sizes[j]),
                   dilation=tuple(int(d) for d in resblock_dilation_sizes[j]), dtype=dtype)
            xs = r if xs is None else xs + r
        x = xs / nk
    x = leaky_relu(x)
    x = conv1d(x, sd["conv_post.weight"], None, padding=3, dtype=dtype)
    return np.tanh(x)


# ------------------------------------------------------------------------------------------------------------
# relative-attention transformer


def mha_rel(sd, x, c, attn_mask_1d=None, *, n_heads, window_size=4, return_attn=False, dtype=np.float64):
    """MultiHeadAttention.forward + attention, rel_transformer.py:138-179.  `attn_mask_1d` is the [B,1,T] (or
    [B,T]) frame mask m; the reference's 4-D mask is m[:, None, :, None] * m[:, None, None, :]
    (RelativeEncoder.forward, rel_transformer.py:291)."""
    q = conv1d(x, sd["conv_q.weight"], sd["conv_q.bias"], dtype=dtype)
    k = conv1d(c, sd["conv_k.weight"], sd["conv_k.bias"], dtype=dtype)
    v = conv1d(c, sd["conv_v.weight"], sd["conv_v.bias"], dtype=dtype)
    q, k, v = _operands(q, k, v)
    B, C, T = q.shape
    dk = C // n_heads
    out = np.empty_like(q)
    p = np.empty((B, n_heads, T, T), dtype=dtype) if return_attn else None
    rel_k = _c(sd["emb_rel_k"], dtype) if window_size is not None else None
    rel_v = _c(sd["emb_rel_v"], dtype) if window_size is not None else None
    m = None if attn_mask_1d is None else _c(np.reshape(attn_mask_1d, (B, T)), dtype)
    _lib(dtype).orc_rel_attention(_p(q), _p(k), _p(v), _p(rel_k), _p(rel_v), _p(m), _p(out), _p(p), _i64(B),
                                  _i64(n_heads), _i64(dk), _i64(T), _i64(-1 if window_size is None else window_size),
                                  _i64(1 if rel_k is None else rel_k.shape[0]))
    y = conv1d(out, sd["conv_o.weight"], sd["conv_o.bias"], dtype=dtype)
    return (y, p) if return_attn else y


def ffn(sd, x, x_mask, *, kernel_size, activation=None, dtype=np.float64):
    """FFN.forward, rel_transformer.py:336-345 (RelativeEncoder builds it without `activation` -> ReLU)."""
    x_mask = _c(x_mask, dtype)
    h = conv1d(_c(x, dtype) * x_mask, sd["conv_1.weight"], sd["conv_1.bias"], padding=kernel_size // 2, dtype=dtype)
    h = h * sigmoid(1.702 * h) if activation == "gelu" else np.maximum(h, 0)
    return conv1d(h * x_mask, sd["conv_2.weight"], sd["conv_2.bias"], dtype=dtype)


def rel_encoder(sd, x, x_mask, g=None, *, n_heads, n_layers, kernel_size, window_size=4, dtype=np.float64):
    """RelativeEncoder.forward (post-LN, pre_ln=False), rel_transformer.py:290-320."""
    x = _c(x, dtype)
    x_mask = _c(x_mask, dtype)
    if g is not None:
        g = conv1d(g, sd["pre_net.weight"], sd["pre_net.bias"], dtype=dtype)
    for i in range(n_layers):
        if g is not None:
            x = x + g
        x = x * x_mask
        y = mha_rel(_sub(sd, f"attn_layers.{i}"), x, x, x_mask, n_heads=n_heads, window_size=window_size, dtype=dtype)
        x = layer_norm_c(x + y, sd[f"norm_layers_1.{i}.gamma"], sd[f"norm_layers_1.{i}.beta"], dtype=dtype)
        y = ffn(_sub(sd, f"ffn_layers.{i}"), x, x_mask, kernel_size=kernel_size, dtype=dtype)
        x = layer_norm_c(x + y, sd[f"norm_layers_2.{i}.gamma"], sd[f"norm_layers_2.{i}.beta"], dtype=dtype)
    return x * x_mask


def frame_prior(sd, x, x_mask, g=None, *, hidden_channels, n_heads, n_layers, kernel_size, dtype=np.float64):
    """FramePriorNetwork.forward, encoder.py:67-73.  NB the reference transposes g (1,2) before the encoder."""
    if g is not None:
        g = np.transpose(g, (0, 2, 1))
    h = rel_encoder(_sub(sd, "encoder"), x, x_mask, g, n_heads=n_heads, n_layers=n_layers, kernel_size=kernel_size,
                    dtype=dtype)
    st = conv1d(h, sd["proj.weight"], sd["proj.bias"], dtype=dtype) * _c(x_mask, dtype)
    return st[:, :hidden_channels], st[:, hidden_channels:]


def pitch_predictor(sd, x, x_mask, spk_emb, *, n_heads, n_layers, kernel_size, dtype=np.float64):
    """PitchPredictor.forward, predictor.py:16-19 -> [B, T, out_dim]."""
    h = rel_encoder(_sub(sd, "pitch_predictor"), x, x_mask, spk_emb, n_heads=n_heads, n_layers=n_layers,
                    kernel_size=kernel_size, dtype=dtype)
    return np.transpose(conv1d(h, sd["linear.weight"], sd["linear.bias"], dtype=dtype), (0, 2, 1))


def phoneme_predictor(sd, x, x_mask, *, n_heads, n_layers, kernel_size, dtype=np.float64):
    """PhonemePredictor.forward, predictor.py:31-35 (log-softmax over the dict dim)."""
    h = rel_encoder(_sub(sd, "phoneme_predictor"), x, x_mask, None, n_heads=n_heads, n_layers=n_layers,
                    kernel_size=kernel_size, dtype=dtype)
    lg = conv1d(h, sd["ph_proj.weight"], sd["ph_proj.bias"], dtype=dtype)
    lg = lg - lg.max(axis=1, keepdims=True)
    return lg - np.log(np.exp(lg).sum(axis=1, keepdims=True))


# ------------------------------------------------------------------------------------------------------------
# integer frame bookkeeping (bit-exact) + positional table


def expand_states(h, mel2token):
    """models/commons/align_ops.py:22-26: prepend a zero row, gather rows by the 1-based index."""
    h = np.asarray(h)
    hp = np.concatenate([np.zeros_like(h[:, :1]), h], axis=1)
    idx = np.asarray(mel2token, dtype=np.int64)
    return np.take_along_axis(hp, idx[..., None].repeat(h.shape[-1], -1), axis=1)


def make_positions(x, padding_idx=0):
    """rel_transformer.py:78-88: cumsum(x != pad) * (x != pad) + pad, int64."""
    mask = (np.asarray(x) != padding_idx).astype(np.int32)
    return (np.cumsum(mask, axis=1).astype(np.int32) * mask).astype(np.int64) + padding_idx


def sinusoid_table(num_embeddings, embedding_dim, padding_idx=None):
    """SinusoidalPositionalEmbedding.get_embedding, rel_transformer.py:59-76 (sin || cos layout), fp32."""
    half = embedding_dim // 2
    e = np.float32(np.log(10000.0) / (half - 1))
    freq = np.exp(np.arange(half, dtype=np.float32) * -e).astype(np.float32)
    ang = np.arange(num_embeddings, dtype=np.float32)[:, None] * freq[None, :]
    emb = np.concatenate([np.sin(ang), np.cos(ang)], axis=1).astype(np.float32)
    if embedding_dim % 2 == 1:
        emb = np.concatenate([emb, np.zeros((num_embeddings, 1), np.float32)], axis=1)
    if padding_idx is not None:
        emb[padding_idx, :] = 0
    return emb


def sinusoidal_positional_embedding(x, embedding_dim, padding_idx=0, init_size=1024):
    """SinusoidalPositionalEmbedding.forward, rel_transformer.py:90-100 -> [B, T, D] (table rows by position)."""
    B, T = x.shape
    n = max(init_size, padding_idx + 1 + T)
    tab = sinusoid_table(n, embedding_dim, padding_idx)
    return tab[make_positions(x, padding_idx).reshape(-1)].reshape(B, T, -1)


def slice_segments(x, ids_str, segment_size=4):
    """modules/commons/utils.py:86-92"""
    x = np.asarray(x)
    return np.stack([x[i, :, int(s):int(s) + segment_size] for i, s in enumerate(ids_str)], 0)


def mel2token_to_dur(mel2token, T_txt, max_dur=None):
    """utils/audio/align.py:105-129: dur[b, i-1] = #{t: mel2token[b, t] == i} for i = 1..T_txt (index 0 dropped)."""
    m = np.asarray(mel2token, dtype=np.int64)
    dur = np.zeros((m.shape[0], T_txt + 1), dtype=np.int64)
    for b in range(m.shape[0]):
        np.add.at(dur[b], m[b], 1)
    dur = dur[:, 1:]
    return dur if max_dur is None else np.minimum(dur, max_dur)


def rand_slice_ids(u, t_len, segment_size):
    """modules/commons/utils.py:95-99: ids = (u * (t_len - segment + 1)).long() with u ~ U[0,1) fp32 from the
    CPU generator; the product is taken in fp32 then truncated."""
    return (np.asarray(u, np.float32) * np.float32(t_len - segment_size + 1)).astype(np.int64)


# ------------------------------------------------------------------------------------------------------------
# text encoder + model glue (inference body)


def text_encoder(sd, text, pitch, dur, mel2ph, *, hidden_channels, n_heads, n_layers, kernel_size, dtype=np.float64):
    """TextEncoder.forward (use_pos_embed=True), encoder.py:34-55."""
    H = hidden_channels
    nonpad = (np.asarray(text) > 0).astype(dtype)[:, None, :]                       # [B,1,Tph]
    sc = np.dtype(dtype).type(np.sqrt(H))                                            # embed_scale, encoder.py:25
    emb = np.concatenate([_c(sd["ph_emb.weight"], dtype)[text] * sc, _c(sd["pitch_emb.weight"], dtype)[pitch] * sc,
                          _c(sd["dur_emb.weight"], dtype)[dur] * sc], axis=2)       # [B,Tph,3H]
    tok = (emb @ _c(sd["linear.weight"], dtype).T + _c(sd["linear.bias"], dtype)) * nonpad.transpose(0, 2, 1)
    # encoder.py:52-54 passes seq_len = token_emb.shape[2] (= H, not T_ph), so the gathered [B*T_ph, H] rows are
    # VIEWED as [B, H, T_ph] and then transposed -- a scramble of the table rows that only type-checks because the
    # element counts agree.  Restated literally: reshape (not transpose) to [B, H, T_ph], then swap axes.
    pos = sinusoidal_positional_embedding(tok[..., 0], H, 0, init_size=max(2000, 1 + H)).astype(dtype)
    pos = pos.reshape(tok.shape[0], H, -1).transpose(0, 2, 1)
    tok = (tok + pos) * nonpad.transpose(0, 2, 1)
    enc = rel_encoder(_sub(sd, "text_encoder"), tok.transpose(0, 2, 1), nonpad, None, n_heads=n_heads,
                      n_layers=n_layers, kernel_size=kernel_size, dtype=dtype)
    out = expand_states(enc.transpose(0, 2, 1), mel2ph)                             # [B,Tmel,H]
    return out.transpose(0, 2, 1)


def forward_pitch(sd, hp, prior, nonpad, spk, voiced_hint=None, hint_tol=0.0, dtype=np.float64):
    """VISinger.forward_pitch with f0=None (synthesis), models/visinger.py:122-135: f0 = pred[..., 0], voiced = pred[..., 1] <= 0,
    condition (f0 * voiced)[:, None, :] * nonpadding -> ([B, 1, T], pred [B, T, 2], voiced [B, T]).
    The voicing decision is a threshold on a computed value: where the oracle's own |pred[..., 1]| <= hint_tol the caller's
    `voiced_hint` decides (a parity test passes the device's decision for exactly those frames and asserts agreement elsewhere)."""
    pred = pitch_predictor(_sub(sd, "pitch_predictor"), prior, nonpad, spk, n_heads=hp["num_heads"],
                           n_layers=hp["pitch_predictor_layers"], kernel_size=hp["ffn_kernel_size"], dtype=dtype)
    voiced = pred[:, :, 1] <= 0
    if voiced_hint is not None:
        near = np.abs(pred[:, :, 1]) <= hint_tol
        voiced = np.where(near, np.asarray(voiced_hint, bool), voiced)
    cond = (pred[:, :, 0] * voiced)[:, None, :] * nonpad
    return cond, pred, voiced


def visinger_infer(sd, hp, text, pitch, dur, mel2ph, spk_id, noise, dtype=np.float64, return_all=False, voiced_hint=None,
                   hint_tol=0.0):
    """VISinger.forward(infer=True), models/visinger.py:71-112.  With hp["use_pitch_embed"] the pitch predictor conditions the frame
    prior (visinger.py:86-90, 122-135).  The reference's own caller cannot run that branch: forward_pitch returns [B, 1, T] and
    FramePriorNetwork.forward transposes it once more before a Conv1d(1, H, 1) (encoder.py:68-69; SURVEY.md 3.5-1: RuntimeError), so the
    branch is restated as the modules define it -- pre_net sees the [B, 1, T] condition -- and is pinned at module level only
    (tests/golden/pitch_predictor.npz, frame_prior.npz, rel_encoder_g.npz)."""
    H = hp["hidden_size"]
    kw = dict(n_heads=hp["num_heads"], kernel_size=hp["ffn_kernel_size"], dtype=dtype)
    nonpad = (np.asarray(mel2ph) > 0).astype(dtype)[:, None, :]
    prior = text_encoder(_sub(sd, "text_encoder"), text, pitch, dur, mel2ph, hidden_channels=H,
                         n_layers=hp["enc_layers"], **kw) * nonpad
    pos = sinusoidal_positional_embedding(prior.transpose(0, 2, 1)[..., 0], H, 0, init_size=2000).astype(dtype)
    prior = prior + pos.transpose(0, 2, 1)
    spk = _c(sd["spk_id_proj.weight"], dtype)[spk_id][:, :, None]                  # [B,gin,1]
    cond, f0_pred, voiced = None, None, None
    if hp.get("use_pitch_embed"):
        cond, f0_pred, voiced = forward_pitch(sd, hp, prior, nonpad, spk, voiced_hint, hint_tol, dtype)
        cond = np.transpose(cond, (0, 2, 1))          # frame_prior() transposes it back, as FramePriorNetwork.forward does
    mu_p, logs_p = frame_prior(_sub(sd, "frame_prior"), prior, nonpad, cond, hidden_channels=H,
                               n_layers=hp["frame_prior_layers"], **kw)
    z_p = (mu_p + _c(noise, dtype) * np.exp(logs_p)) * nonpad
    z_q = flow_block(_sub(sd, "flow"), z_p, nonpad, spk, reverse=True, channels=H, hidden_channels=H, kernel_size=5,
                     dilation_rate=1, n_layers=4, dtype=dtype) * nonpad
    wav = generator(_sub(sd, "decoder"), z_q * nonpad, spk, resblock=hp["dec_blocks"],
                    resblock_kernel_sizes=hp["dec_kernel_size"], resblock_dilation_sizes=hp["dec_dilation_sizes"],
                    upsample_rates=hp["upsample_rates"], upsample_kernel_sizes=hp["upsample_kernel_sizes"],
                    dtype=dtype)[:, 0]
    if return_all:
        return dict(prior=prior, mu_p=mu_p, logs_p=logs_p, z_p=z_p, z_q=z_q, wav_out=wav, f0_pred=f0_pred, voiced=voiced)
    return wav


# ------------------------------------------------------------------------------------------------------------
# discriminators (a13; BASELINE config 3 only)


def discriminator_s(sd, x, dtype=np.float64):
    """DiscriminatorS.forward, modules/discriminator.py:64-75 -> (flat logits, fmap list)."""
    cfg = [(15, 1, 7, 1), (41, 4, 20, 4), (41, 4, 20, 16), (41, 4, 20, 64), (41, 4, 20, 256), (5, 1, 2, 1)]
    fmap = []
    x = _c(x, dtype)
    for i, (k, st, pad, g) in enumerate(cfg):
        x = leaky_relu(conv1d_sg(x, _get_w(sd, f"convs.{i}", dtype), sd[f"convs.{i}.bias"], st, pad, g, dtype=dtype))
        fmap.append(x)
    x = conv1d_sg(x, _get_w(sd, "conv_post", dtype), sd["conv_post.bias"], 1, 1, 1, dtype=dtype)
    fmap.append(x)
    return x.reshape(x.shape[0], -1), fmap


def discriminator_p(sd, x, period, kernel_size=5, stride=3, dtype=np.float64):
    """DiscriminatorP.forward, modules/discriminator.py:28-47: reflect-pad to a multiple of the period, view as
    [B, 1, T/p, p], (k,1) convs along the first axis -> per column j a strided conv1d over x[:, :, j::p]."""
    x = _c(x, dtype)
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = np.pad(x, ((0, 0), (0, 0), (0, n_pad)), mode="reflect")
        t += n_pad
    x = x.reshape(b, c, t // period, period)
    pad = get_padding(kernel_size, 1)

    def conv2d_k1(x4, w, bias, st, pd):
        B_, C_, H_, W_ = x4.shape
        cols = np.ascontiguousarray(x4.transpose(0, 3, 1, 2)).reshape(B_ * W_, C_, H_)     # each column is a sequence
        y = conv1d_sg(cols, w[:, :, :, 0], bias, st, pd, 1, dtype=dtype)
        return np.ascontiguousarray(y.reshape(B_, W_, y.shape[1], y.shape[2]).transpose(0, 2, 3, 1))

    fmap = []
    for i in range(5):
        st = stride if i < 4 else 1
        x = leaky_relu(conv2d_k1(x, _get_w(sd, f"convs.{i}", dtype), sd[f"convs.{i}.bias"], st, pad))
        fmap.append(x)
    x = conv2d_k1(x, _get_w(sd, "conv_post", dtype), sd["conv_post.bias"], 1, 1)
    fmap.append(x)
    return x.reshape(b, -1), fmap


# ---------------------------------------------------------------------------------------------------------------------
# linear / log-mel spectrogram in fp64, by definition (framed DFT with numpy's rfft in double).
# PARITY UNPINNED w.r.t. the reference: utils/audio/mel_processing.py:15-38 delegates to torchaudio.transforms
# (Spectrogram / MelSpectrogram), a third-party dependency that is neither under /root/reference nor installed here
# (requirements.txt:17 leaves it unpinned; README.md:22 suggests torchaudio==0.11.0).  What follows restates the documented
# defaults of that version: periodic hann window of win_length centred in n_fft, center=True with reflect padding of n_fft/2,
# onesided, power 2, no normalisation; melscale_fbanks(mel_scale="htk", norm=None); the reference's own additions are
# log(x + 1e-3) and dropping the last frame (mel_processing.py:24-38).  It referees visinger_amd/audio.py's torch.stft path.
def linear_spectrogram_f64(wav, n_fft=2048, win_length=1200, hop_length=300, power=2.0):
    """wav [B, L] -> [B, T, n_fft/2 + 1] (float64), T = L // hop_length (last of the 1 + L // hop frames dropped)"""
    wav = np.asarray(wav, np.float64)
    B, Lw = wav.shape
    n = np.arange(win_length)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)                  # torch.hann_window(periodic=True)
    full = np.zeros(n_fft)
    left = (n_fft - win_length) // 2
    full[left:left + win_length] = win
    x = np.pad(wav, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
    n_frames = 1 + Lw // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    spec = np.fft.rfft(x[:, idx] * full, axis=-1)                           # [B, frames, n_fft/2+1]
    return (np.abs(spec) ** power)[:, :-1]


def mel_filterbank_f64(n_freqs, f_min, f_max, n_mels, sample_rate):
    """HTK mel triangles without area normalisation, [n_freqs, n_mels] (float64)"""
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + np.asarray(f, np.float64) / 700.0)      # noqa: E731
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def mel_spectrogram_f64(wav, sample_rate=24000, n_fft=2048, win_length=1200, hop_length=300, n_mels=128, f_min=20.0, f_max=12000.0,
                        eps=1e-3):
    """wav [B, L] -> log-mel [B, T, n_mels] (float64)"""
    lin = linear_spectrogram_f64(wav, n_fft, win_length, hop_length)
    return np.log(lin @ mel_filterbank_f64(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate) + eps)
