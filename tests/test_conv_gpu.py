"""GPU parity of the conv engine (C ABI vs_conv_*) against the CPU oracle, op by op.
Tolerance: fp32 MFMA accumulation vs the fp64 oracle -> |err| <= 2e-5 * (1 + |ref|) at these reduction lengths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from visinger_amd import _lib as L  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(got, ref, tol=2e-5):
    got = got.detach().cpu().double().numpy()
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = np.abs(got - ref) / (1.0 + np.abs(ref))
    assert np.isfinite(got).all()
    assert err.max() <= tol, f"max scaled err {err.max():.3e}"


def rng(seed):
    return np.random.default_rng(seed)


CONV_CASES = [
    # (B, Cin, Cout, T, k, dil)
    (2, 16, 32, 37, 5, 1),
    (1, 1, 16, 50, 1, 1),        # pre_net Conv1d(1, H, 1)
    (2, 96, 192, 300, 1, 1),     # coupling pre
    (2, 192, 96, 300, 1, 1),     # coupling post (3 tiles)
    (1, 128, 128, 700, 11, 5),   # resblock k11 d5 (span 50), ragged tile edge
    (1, 64, 64, 1100, 7, 3),
    (2, 32, 32, 1300, 3, 1),
    (1, 32, 1, 900, 7, 1),       # conv_post shape
    (1, 192, 512, 64, 7, 1),     # conv_pre
    (1, 33, 70, 129, 9, 1),      # odd sizes (channel padding, 3 tiles with tail)
    (1, 256, 256, 260, 3, 3),
]


@pytest.mark.parametrize("B,Cin,Cout,T,k,dil", CONV_CASES)
def test_conv1d_plain(oracle, B, Cin, Cout, T, k, dil):
    from visinger_amd.ops import ConvOp
    r = rng(B * 1000 + Cin + Cout + T + k)
    x = r.standard_normal((B, Cin, T)).astype(np.float32)
    w = (r.standard_normal((Cout, Cin, k)) / np.sqrt(Cin * k)).astype(np.float32)
    bias = r.standard_normal(Cout).astype(np.float32)
    pad = (k * dil - dil) // 2
    ref = oracle.conv1d(x, w, bias, dilation=dil, padding=pad)
    op = ConvOp(L.CONV1D, Cin, Cout, k, dil, pad)
    op.set_weights(dev(w), None, dev(bias))
    y = op.forward(dev(x))
    torch.cuda.synchronize()
    close(y, ref)


def test_conv1d_weightnorm_lrelu_residual_acc_scale(oracle):
    from visinger_amd.ops import ConvOp, weightnorm_fold
    r = rng(7)
    B, C, T, k, dil = 2, 64, 333, 7, 3
    x = r.standard_normal((B, C, T)).astype(np.float32)
    v = r.standard_normal((C, C, k)).astype(np.float32)
    g = (0.5 + r.random((C, 1, 1))).astype(np.float32)
    bias = r.standard_normal(C).astype(np.float32)
    res = r.standard_normal((B, C, T)).astype(np.float32)
    accb = r.standard_normal((B, C, T)).astype(np.float32)
    w = oracle.weight_norm(v, g)
    close(weightnorm_fold(dev(v), dev(g)), w, tol=2e-6)
    pad = (k * dil - dil) // 2
    ref = (oracle.conv1d(oracle.leaky_relu(x.astype(np.float64)), w, bias, dilation=dil, padding=pad) + res + accb) / 3.0
    op = ConvOp(L.CONV1D, C, C, k, dil, pad)
    op.set_weights(dev(v), dev(g), dev(bias))
    y = op.forward(dev(x), in_act=L.IN_LRELU, res=dev(res), acc=dev(accb), scale=1.0 / 3.0)
    close(y, ref)
    # in-place accumulate (y aliases acc) and tanh
    acc_t = dev(accb)
    op.forward(dev(x), in_act=L.IN_LRELU, y=acc_t, acc=acc_t, out_act=L.OUT_TANH)
    ref2 = np.tanh(oracle.conv1d(oracle.leaky_relu(x.astype(np.float64)), w, bias, dilation=dil, padding=pad) + accb)
    close(acc_t, ref2)


def test_conv1d_mask_in_out_relu_and_bias_b(oracle):
    from visinger_amd.ops import ConvOp
    r = rng(8)
    B, Cin, Cout, T, k = 2, 48, 96, 211, 9
    x = r.standard_normal((B, Cin, T)).astype(np.float32)
    w = (r.standard_normal((Cout, Cin, k)) / np.sqrt(Cin * k)).astype(np.float32)
    bias = r.standard_normal(Cout).astype(np.float32)
    bb = r.standard_normal((B, Cout)).astype(np.float32)
    mask = np.zeros((B, T), np.float32)
    mask[0, :T] = 1
    mask[1, :140] = 1
    ref = oracle.conv1d(x * mask[:, None], w, bias, padding=k // 2) + bb[:, :, None]
    ref = np.maximum(ref, 0) * mask[:, None]
    op = ConvOp(L.CONV1D, Cin, Cout, k, 1, k // 2)
    op.set_weights(dev(w), None, dev(bias))
    y = op.forward(dev(x), in_act=L.IN_MASK, mask=dev(mask), bias_b=dev(bb), out_act=L.OUT_RELU, out_mask=True)
    close(y, ref)


@pytest.mark.parametrize("Cin,Cout,k,u,T", [(64, 32, 16, 8, 40), (32, 32, 4, 2, 300), (48, 24, 11, 5, 33),
                                             (32, 16, 7, 3, 50), (512, 256, 16, 8, 16), (10, 6, 8, 4, 13),
                                             (48, 32, 11, 5, 159), (128, 64, 7, 3, 40)])     # padding row tiles + > 2 chunks
def test_conv_transpose1d(oracle, Cin, Cout, k, u, T):
    from visinger_amd.ops import ConvOp
    r = rng(Cin + Cout + k + u)
    B = 2
    x = r.standard_normal((B, Cin, T)).astype(np.float32)
    v = r.standard_normal((Cin, Cout, k)).astype(np.float32)
    g = (0.5 + r.random((Cin, 1, 1))).astype(np.float32)
    bias = r.standard_normal(Cout).astype(np.float32)
    pad = (k - u) // 2
    w = oracle.weight_norm(v, g)                       # norm over dim 0 = input channel
    ref = oracle.conv_transpose1d(oracle.leaky_relu(x.astype(np.float64)), w, bias, stride=u, padding=pad)
    op = ConvOp(L.CONV_TRANSPOSE1D, Cin, Cout, k, u, pad)
    op.set_weights(dev(v), dev(g), dev(bias))
    y = op.forward(dev(x), in_act=L.IN_LRELU)
    assert y.shape[-1] == ref.shape[-1]
    close(y, ref)


@pytest.mark.parametrize("H,Cin,T,k,dil", [(192, 192, 300, 5, 1), (16, 16, 37, 5, 1), (40, 24, 70, 3, 2)])
def test_paired_gate(oracle, H, Cin, T, k, dil):
    """WaveNet in_layer + conditioning + tanh*sigmoid gate (encoder.py:175-185,206-213)."""
    from visinger_amd.ops import ConvOp
    r = rng(H + T)
    B = 2
    x = r.standard_normal((B, Cin, T)).astype(np.float32)
    w = (r.standard_normal((2 * H, Cin, k)) / np.sqrt(Cin * k)).astype(np.float32)
    bias = r.standard_normal(2 * H).astype(np.float32)
    gl = r.standard_normal((B, 2 * H)).astype(np.float32)
    pad = (k * dil - dil) // 2
    pre = oracle.conv1d(x, w, bias, dilation=dil, padding=pad) + gl[:, :, None]
    ref = np.tanh(pre[:, :H]) * oracle.sigmoid(pre[:, H:])
    op = ConvOp(L.CONV1D_PAIRED, Cin, 2 * H, k, dil, pad)
    op.set_weights(dev(w), None, dev(bias))
    y = op.forward(dev(x), bias_b=dev(gl), pair_mode=L.PAIR_GATE)
    assert y.shape == (B, H, T)
    close(y, ref)


@pytest.mark.parametrize("H,T,split32", [(192, 300, True), (16, 37, False)])
def test_res_skip_split(oracle, H, T, split32):
    """WaveNet res/skip 1x1 conv with the two destinations of encoder.py:188-192."""
    from visinger_amd.ops import ConvOp
    r = rng(H)
    B = 2
    acts = r.standard_normal((B, H, T)).astype(np.float32)
    x = r.standard_normal((B, H, T)).astype(np.float32)
    out = r.standard_normal((B, H, T)).astype(np.float32)
    w = (r.standard_normal((2 * H, H, 1)) / np.sqrt(H)).astype(np.float32)
    bias = r.standard_normal(2 * H).astype(np.float32)
    mask = np.ones((B, T), np.float32)
    mask[1, T - 9:] = 0
    rs = oracle.conv1d(acts, w, bias)
    ref_x = (x + rs[:, :H]) * mask[:, None]
    ref_out = out + rs[:, H:]
    op = ConvOp(L.CONV1D, H, 2 * H, 1, 1, 0)
    op.set_weights(dev(w), None, dev(bias))
    xt, ot = dev(x), dev(out)
    op.forward(dev(acts), y=xt, res=xt, out_mask=True, mask=dev(mask), split_row=H, out1=dict(y=ot, acc=ot))
    close(xt, ref_x)
    close(ot, ref_out)


@pytest.mark.parametrize("mean_only", [True, False])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("flip", [False, True])
def test_coupling_post_epilogue(oracle, mean_only, reverse, flip):
    """post conv + affine coupling update on the x1 half, in place, with the Flip folded into the weights
    (flow.py:70-85,88-95)."""
    from visinger_amd.ops import ConvOp
    r = rng(11 + mean_only + 2 * reverse)
    B, H, half, T = 2, 64, 48, 150
    C = 2 * half
    h = r.standard_normal((B, H, T)).astype(np.float32)
    xfull = r.standard_normal((B, C, T)).astype(np.float32)
    nrow = half if mean_only else 2 * half
    w = (0.5 * r.standard_normal((nrow, H, 1)) / np.sqrt(H)).astype(np.float32)
    bias = (0.1 * r.standard_normal(nrow)).astype(np.float32)
    mask = np.ones((B, T), np.float32)
    mask[0, 100:] = 0
    stats = oracle.conv1d(h, w, bias) * mask[:, None]
    m = stats[:, :half]
    logs = np.zeros_like(m) if mean_only else stats[:, half:]
    # logical view: with flip the physical tensor holds the channels reversed
    xlog = xfull[:, ::-1] if flip else xfull
    x1 = xlog[:, half:]
    if not reverse:
        x1n = m + x1 * np.exp(logs) * mask[:, None]
        ref_ld = logs.sum(axis=(1, 2))
    else:
        x1n = (x1 - m) * np.exp(-logs) * mask[:, None]
    ref_log = np.concatenate([xlog[:, :half], x1n], 1)
    ref_phys = ref_log[:, ::-1] if flip else ref_log
    xt = dev(xfull)
    # physical location of logical x1: rows [half, C) without flip, rows [0, half) reversed with flip
    row0 = 0 if flip else half
    flags = L.FLIP_OUT if flip else 0
    kind = L.CONV1D if mean_only else L.CONV1D_PAIRED
    op = ConvOp(kind, H, nrow, 1, 1, 0, flags)
    op.set_weights(dev(w), None, dev(bias))
    from visinger_amd.ops import _off
    x1p = _off(xt, row0 * T)
    ld = torch.zeros(B, device="cuda")
    if mean_only:
        op.forward(dev(h), mask=dev(mask), y_ptr=x1p, res_ptr=x1p, y_bs=C * T, res_bs=C * T,
                   mode=L.MODE_COUPLING_MEAN_INV if reverse else L.MODE_COUPLING_MEAN_FWD)
    else:
        op.forward(dev(h), mask=dev(mask), y_ptr=x1p, res_ptr=x1p, y_bs=C * T, res_bs=C * T,
                   pair_mode=L.PAIR_COUPLING_INV if reverse else L.PAIR_COUPLING_FWD, logdet=ld)
    close(xt, ref_phys)
    if not reverse and not mean_only:
        got = ld.cpu().double().numpy()
        assert np.abs(got - ref_ld).max() <= 1e-4 * np.abs(ref_ld).max()


def test_flip_in_and_channel_window(oracle):
    """pre conv reading the logical x0 half from the physical tensor (upper half, reversed) -- flow.py:67-68."""
    from visinger_amd.ops import ConvOp, _off
    r = rng(21)
    B, half, H, T = 2, 48, 64, 130
    C = 2 * half
    xfull = r.standard_normal((B, C, T)).astype(np.float32)
    w = (r.standard_normal((H, half, 1)) / np.sqrt(half)).astype(np.float32)
    bias = r.standard_normal(H).astype(np.float32)
    mask = np.ones((B, T), np.float32)
    mask[1, 77:] = 0
    xlog = xfull[:, ::-1]
    ref = oracle.conv1d(np.ascontiguousarray(xlog[:, :half]), w, bias) * mask[:, None]
    op = ConvOp(L.CONV1D, half, H, 1, 1, 0, L.FLIP_IN)
    op.set_weights(dev(w), None, dev(bias))
    xt = dev(xfull)
    y = torch.empty(B, H, T, device="cuda")
    op.forward(None, B=B, T=T, x_ptr=_off(xt, half * T), x_bs=C * T, y=y, mask=dev(mask), out_mask=True)
    close(y, ref)


def test_errors_are_reported_not_fatal():
    from visinger_amd.ops import ConvOp
    with pytest.raises(L.VisingerHipError):
        ConvOp(L.CONV1D, 16, 16, 33, 5, 80)       # span 160 > LDS window
    op = ConvOp(L.CONV1D, 16, 16, 3, 1, 1)
    with pytest.raises(L.VisingerHipError):
        op.forward(torch.zeros(1, 16, 8, device="cuda"))   # weights not set


# F(2,3) minimal-filtering path (conv_wino_kernel): every (k, dilation) of the resblocks + the FFN k=9, the three
# workgroup shapes (C_out % 128 == 0, % 64 == 0, odd tile counts), aligned lengths (vector epilogue, interior + ragged
# last tile) and unaligned lengths (element-wise epilogue everywhere), with every fused epilogue option.
WINO_CASES = [
    # (B, Cin, Cout, T, k, dil)
    (2, 128, 128, 1024, 3, 1), (1, 128, 128, 1000, 3, 3), (1, 128, 128, 772, 3, 5),
    (1, 64, 64, 2048, 7, 1), (2, 64, 64, 1300, 7, 3), (1, 64, 64, 1504, 7, 5),
    (1, 32, 32, 4096, 11, 1), (1, 32, 32, 2600, 11, 3), (2, 32, 32, 2500, 11, 5),
    (1, 256, 256, 520, 3, 1), (1, 192, 768, 300, 9, 1), (1, 768, 192, 300, 9, 1),
    (1, 40, 96, 333, 5, 1), (1, 32, 32, 7, 3, 1), (1, 128, 128, 129, 7, 5),
]


@pytest.mark.parametrize("B,Cin,Cout,T,k,dil", WINO_CASES)
def test_conv1d_winograd_path(oracle, vs_option, B, Cin, Cout, T, k, dil):
    from visinger_amd.ops import ConvOp
    vs_option("VS_WINO_FORCE", 1)     # also the shapes the dispatch heuristic leaves on the direct engine
    r = rng(B * 977 + Cin + 3 * Cout + T + 11 * k + dil)
    x = r.standard_normal((B, Cin, T)).astype(np.float32)
    v = r.standard_normal((Cout, Cin, k)).astype(np.float32)
    g = (0.5 + r.random((Cout, 1, 1))).astype(np.float32)
    bias = r.standard_normal(Cout).astype(np.float32)
    res = r.standard_normal((B, Cout, T)).astype(np.float32)
    accb = r.standard_normal((B, Cout, T)).astype(np.float32)
    mask = np.ones((B, T), np.float32)
    mask[-1, (2 * T) // 3:] = 0
    w = oracle.weight_norm(v, g)
    pad = (k * dil - dil) // 2
    op = ConvOp(L.CONV1D, Cin, Cout, k, dil, pad).set_math(L.MATH_F32)
    op.set_weights(dev(v), dev(g), dev(bias))
    conv = oracle.conv1d(oracle.leaky_relu(x.astype(np.float64)), w, bias, dilation=dil, padding=pad)
    y = op.forward(dev(x), in_act=L.IN_LRELU)
    assert op.kernel_instance().startswith("conv_wino_kernel"), op.kernel_instance()      # as reported by the library
    close(y, conv)
    y = op.forward(dev(x), in_act=L.IN_LRELU, res=dev(res), acc=dev(accb), scale=1.0 / 3.0)
    close(y, (conv + res + accb) / 3.0)
    convm = oracle.conv1d(oracle.leaky_relu(x.astype(np.float64)) * mask[:, None], w, bias, dilation=dil, padding=pad)
    y = op.forward(dev(x), in_act=L.IN_LRELU_MASK, mask=dev(mask), res=dev(res), out_act=L.OUT_TANH, out_mask=True)
    close(y, np.tanh(convm + res) * mask[:, None])


def test_winograd_matches_direct_engine_at_size(vs_option):
    """size-independent property at a resblock-sized launch: the F(2,3) path and the direct path agree to fp32 rounding"""
    from visinger_amd.ops import ConvOp
    torch.manual_seed(3)
    B, C, T, k, dil = 4, 128, 16384, 7, 3
    x = torch.randn(B, C, T, device="cuda")
    w = torch.randn(C, C, k, device="cuda") / (C * k) ** 0.5
    bias = torch.randn(C, device="cuda")
    op = ConvOp(L.CONV1D, C, C, k, dil, (k * dil - dil) // 2).set_math(L.MATH_F32)
    op.set_weights(w, None, bias)
    vs_option("VS_WINO_FORCE", 1)
    y_w = op.forward(x, in_act=L.IN_LRELU, res=x)
    vs_option("VS_WINO_FORCE", 0)
    vs_option("VS_NO_WINO", 1)
    y_d = op.forward(x, in_act=L.IN_LRELU, res=x)
    torch.cuda.synchronize()
    assert float((y_w - y_d).abs().max()) <= 2e-5 * (1.0 + float(y_d.abs().max()))


@pytest.mark.parametrize("seed", [11, 12])
def test_conv_engine_random_sweep(seed, monkeypatch, vs_option):
    """tools/conv_fuzz.py: 150 random (shape, dilation / stride, fused option) cases per seed across every kernel instance"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import conv_fuzz
    monkeypatch.setattr(sys, "argv", ["conv_fuzz.py", "150", str(seed)])
    vs_option("VS_WINO_FORCE", 0)
    conv_fuzz.main()          # (sets VS_WINO_FORCE per case through L.set_option; the fixture restores it)


@pytest.mark.parametrize("C,k,d,T,B", [(32, 3, 1, 2048, 2), (32, 7, 3, 1500, 1), (32, 11, 5, 1024, 2), (64, 3, 5, 1000, 1),
                                        (64, 7, 1, 700, 2), (64, 11, 3, 516, 1), (32, 5, 1, 37, 1), (32, 3, 3, 4, 2), (64, 9, 1, 250, 1)])
@pytest.mark.parametrize("math", [L.MATH_SPLIT6, L.MATH_F32, L.MATH_BF16])
def test_resblock_pair_fused_launch(oracle, C, k, d, T, B, math):
    """csrc/resblock_pair_split.hip (split-bf16 x6, the default; plain bf16 operands for VS_MATH_BF16, at a bf16 tolerance: two
    convs of C*k bf16 products each) and csrc/resblock_pair.hip (fp32 MFMA): y = conv2(lrelu(conv1(lrelu(x)))) + x [+ acc] [* scale] in one launch vs the fp64 oracle; tiles
    in the interior (vector epilogue), at both sequence ends, lengths below one tile and not a multiple of 4 (element-wise
    epilogue), every (k, dilation) of the MRF blocks."""
    from visinger_amd.ops import ConvOp, respair_forward, respair_supported
    r = rng(C * 31 + k * 7 + d + T)
    x = r.standard_normal((B, C, T)).astype(np.float32)
    w1 = (r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    w2 = (r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    b1, b2 = r.standard_normal(C).astype(np.float32), r.standard_normal(C).astype(np.float32)
    accb = r.standard_normal((B, C, T)).astype(np.float32)
    op1 = ConvOp(L.CONV1D, C, C, k, d, d * (k - 1) // 2).set_math(math)
    op2 = ConvOp(L.CONV1D, C, C, k, 1, (k - 1) // 2).set_math(math)
    op1.set_weights(dev(w1), None, dev(b1))
    op2.set_weights(dev(w2), None, dev(b2))
    assert respair_supported(op1, op2)
    t = oracle.conv1d(oracle.leaky_relu(x.astype(np.float64)), w1, b1, dilation=d, padding=d * (k - 1) // 2)
    ref = oracle.conv1d(oracle.leaky_relu(t), w2, b2, padding=(k - 1) // 2) + x
    xd = dev(x)
    tol = 2e-5 if math != L.MATH_BF16 else 3e-2
    y = respair_forward(op1, op2, xd, torch.empty_like(xd), res=xd)
    close(y, ref, tol)
    assert op1.kernel_instance().startswith("respair_kernel<" if math == L.MATH_F32 else "respair_split_kernel<")
    assert op1.kernel_instance().endswith({L.MATH_F32: ">", L.MATH_SPLIT6: ", 6, false>", L.MATH_BF16: ", 1, false>"}[math])
    acc_t = dev(accb)
    respair_forward(op1, op2, xd, acc_t, res=xd, acc=acc_t, scale=1.0 / 3.0)       # in-place accumulate, MRF average
    close(acc_t, (ref + accb) / 3.0, tol)
