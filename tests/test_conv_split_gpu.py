"""The split-bf16 conv engine (csrc/conv_split.hip, vs_conv_set_math) through the C ABI: parity with the fp64 oracle for every
conv kind and fused epilogue, error no larger than the fp32 MFMA engine's, bf16 mode within a bf16 tolerance, and re-packing
when the arithmetic of a live handle changes.  (Reference sites: the nn.Conv1d / ConvTranspose1d calls listed in
include/visinger_hip.h; the oracle restates torch's conv arithmetic in fp64.)"""
import numpy as np
import pytest
import torch

from visinger_amd import _lib as L

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def rel_rms(y, ref):
    e = y.detach().cpu().double().numpy() - ref
    return float(np.sqrt((e ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-30))


@pytest.mark.parametrize("C,k,d,T", [(128, 3, 1, 2048), (128, 7, 3, 1500), (256, 11, 1, 777), (64, 11, 5, 4096), (32, 7, 1, 3000),
                                      (192, 5, 1, 1000), (96, 1, 1, 513)])
def test_split6_error_not_above_fp32_mfma(oracle, vs_option, C, k, d, T):
    """the six-product split keeps the fp32 class: its RMS error against fp64 is not above the exact-fp32 MFMA engine's
    direct form (both are dominated by fp32 accumulation rounding; tools/conv_accuracy.py prints the table)"""
    from visinger_amd.ops import ConvOp
    vs_option("VS_NO_WINO", 1)        # MATH_F32 = the direct fp32 MFMA kernel, not its F(2,3) variant
    r = np.random.default_rng(C * 31 + k * 7 + d + T)
    x = r.standard_normal((2, C, T)).astype(np.float32)
    w = (r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    bias = r.standard_normal(C).astype(np.float32)
    pad = d * (k - 1) // 2
    ref = oracle.conv1d(x.astype(np.float64), w, bias, dilation=d, padding=pad)
    errs = {}
    for math in (L.MATH_F32, L.MATH_SPLIT6, L.MATH_SPLIT3, L.MATH_BF16):
        op = ConvOp(L.CONV1D, C, C, k, d, pad).set_math(math)
        assert op.math == math
        op.set_weights(dev(w), None, dev(bias))
        errs[math] = rel_rms(op.forward(dev(x)), ref)
    assert errs[L.MATH_SPLIT6] <= 1.25 * errs[L.MATH_F32] + 1e-8, errs
    assert errs[L.MATH_SPLIT6] <= 3e-6, errs
    # the split-f16 x3 engine (the default since round 3): 22-bit operands under a per-tile scale, half the products summed in fp32 --
    # not above the fp32 MFMA's error either, and not above the six-product split's
    assert errs[L.MATH_SPLIT3] <= 1.0 * errs[L.MATH_F32] + 1e-8, errs
    assert errs[L.MATH_SPLIT3] <= 1.05 * errs[L.MATH_SPLIT6] + 1e-8, errs
    assert 1e-4 < errs[L.MATH_BF16] <= 6e-3, errs          # bf16 operands: ~2^-9 per product, fp32 accumulate


@pytest.mark.parametrize("Cin,Cout,B,with_bias", [(256, 1536, 32, True), (256, 192, 32, True), (256, 512, 3, False), (200, 70, 40, True), (16, 5, 1, True),
                                                   (512, 64, 33, False)])
def test_single_frame_1x1_conv(oracle, vs_option, Cin, Cout, B, with_bias):
    """conv_t1_kernel (round 6): 1 x 1 convs over ONE frame per item -- the conditioning vectors (WN.cond_layer 256 -> 2 * hidden * n_layers on the speaker
    embedding, the g-convs of the flow and the generator) -- leave the tile kernels (one valid column in 128, 63 us for 256 -> 1536 at B = 32) for a workgroup
    per 32-row tile of W over up to 32 items: against the fp64 oracle in the fp32-class arithmetics, against the oracle on bf16-rounded operands in
    VS_MATH_BF16, and against the tile kernel it replaces (VS_NO_T1_CONV=1); a per-item bias and an output scale ride along; T = 2 stays on the tile kernels."""
    from visinger_amd.ops import ConvOp
    r = np.random.default_rng(Cin + 3 * Cout + B)
    x = r.standard_normal((B, Cin, 1)).astype(np.float32)
    w = (r.standard_normal((Cout, Cin, 1)) / np.sqrt(Cin)).astype(np.float32)
    bias = r.standard_normal(Cout).astype(np.float32) if with_bias else None
    ref = oracle.conv1d(x.astype(np.float64), w, bias)
    for math in (L.MATH_SPLIT3, L.MATH_SPLIT6, L.MATH_BF16):
        op = ConvOp(L.CONV1D, Cin, Cout, 1, 1, 0).set_math(math)
        op.set_weights(dev(w), None, None if bias is None else dev(bias))
        y = op.forward(dev(x))
        assert op.kernel_instance() == "conv_t1_kernel" and y.shape == (B, Cout, 1)
        vs_option("VS_NO_T1_CONV", 1)
        y_tile = op.forward(dev(x))
        assert op.kernel_instance() != "conv_t1_kernel"
        vs_option("VS_NO_T1_CONV", 0)
        if math == L.MATH_BF16:
            with oracle.operand_rounding("bf16"):
                ref16 = oracle.conv1d(x.astype(np.float64), w, bias)
            assert rel_rms(y, ref16) <= 2e-6 and rel_rms(y_tile, ref16) <= 2e-6
        else:
            assert rel_rms(y, ref) <= 4e-7, (math, rel_rms(y, ref))
            assert float((y - y_tile).abs().max()) <= 2e-5
    op = ConvOp(L.CONV1D, Cin, Cout, 1, 1, 0)
    op.set_weights(dev(w), None, None if bias is None else dev(bias))
    bb = r.standard_normal((B, Cout)).astype(np.float32)
    y3 = op.forward(dev(x), bias_b=dev(bb), scale=0.5)
    assert op.kernel_instance() == "conv_t1_kernel" and rel_rms(y3, 0.5 * (ref + bb[:, :, None].astype(np.float64))) <= 3e-7
    x2 = r.standard_normal((B, Cin, 2)).astype(np.float32)
    y2 = op.forward(dev(x2))
    assert op.kernel_instance() != "conv_t1_kernel" and rel_rms(y2, oracle.conv1d(x2.astype(np.float64), w, bias)) <= 3e-6


def test_split_kernel_instances_and_repack(oracle):
    """vs_conv_set_math on a handle whose weights are already packed re-packs the bf16 planes; the three arithmetics of one
    handle agree with the oracle in turn"""
    from visinger_amd.ops import ConvOp
    r = np.random.default_rng(5)
    B, Cin, Cout, T, k = 2, 48, 160, 700, 5
    x = r.standard_normal((B, Cin, T)).astype(np.float32)
    w = (r.standard_normal((Cout, Cin, k)) / np.sqrt(Cin * k)).astype(np.float32)
    ref = oracle.conv1d(oracle.leaky_relu(x.astype(np.float64)), w, None, padding=2)
    op = ConvOp(L.CONV1D, Cin, Cout, k, 1, 2)
    assert op.math == L.MATH_SPLIT3 and op.kernel_instance() == ""          # the library default
    op.set_weights(dev(w), None, None)
    for math, tol in ((L.MATH_SPLIT3, 2e-6), (L.MATH_SPLIT6, 2e-6), (L.MATH_F32, 2e-6), (L.MATH_BF16, 6e-3), (L.MATH_SPLIT3, 2e-6), (L.MATH_SPLIT6, 2e-6)):
        op.set_math(math)
        assert rel_rms(op.forward(dev(x), in_act=L.IN_LRELU), ref) <= tol, math
        # the instance name comes from the library's own dispatch (vs_last_kernel_name), template arguments as rocprofv3 prints them
        # (a 12-workgroup launch like this one takes the 32-row tiles: conv_split_kernel<1, 2, 1, 4, TERMS>)
        name = op.kernel_instance()
        if math == L.MATH_F32:
            assert name.startswith(("conv_wino_kernel<", "conv_mfma_kernel<")), name
        else:
            assert name.startswith("conv_split_kernel<1, ") and name.endswith(f", {math}>"), name
    with pytest.raises(L.VisingerHipError):
        op.set_math(4)


@pytest.mark.parametrize("math,tol", [(L.MATH_SPLIT6, 3e-6), (L.MATH_SPLIT3, 3e-6), (L.MATH_BF16, 8e-3)])
def test_split_transposed_and_paired(oracle, math, tol):
    """polyphase transposed conv, WaveNet gate and affine-coupling epilogues on the split engine"""
    from visinger_amd.ops import ConvOp
    r = np.random.default_rng(17)
    # transposed 64 -> 32, k = 16, stride 8 (decoder.py:26-31)
    x = r.standard_normal((2, 64, 300)).astype(np.float32)
    w = (r.standard_normal((64, 32, 16)) / np.sqrt(64 * 2)).astype(np.float32)
    bias = r.standard_normal(32).astype(np.float32)
    ref = oracle.conv_transpose1d(oracle.leaky_relu(x.astype(np.float64)), w, bias, stride=8, padding=4)
    op = ConvOp(L.CONV_TRANSPOSE1D, 64, 32, 16, 8, 4).set_math(math)
    op.set_weights(dev(w), None, dev(bias))
    assert rel_rms(op.forward(dev(x), in_act=L.IN_LRELU), ref) <= tol
    # gate: tanh(a) * sigmoid(b) over the two halves of a 192 -> 384 k5 conv (wavenet.py:54-64)
    H, T = 192, 500
    x = r.standard_normal((2, H, T)).astype(np.float32)
    w = (r.standard_normal((2 * H, H, 5)) / np.sqrt(H * 5)).astype(np.float32)
    bias = r.standard_normal(2 * H).astype(np.float32)
    z = oracle.conv1d(x.astype(np.float64), w, bias, padding=2)
    ref = np.tanh(z[:, :H]) * (1.0 / (1.0 + np.exp(-z[:, H:])))
    op = ConvOp(L.CONV1D_PAIRED, H, 2 * H, 5, 1, 2).set_math(math)
    op.set_weights(dev(w), None, dev(bias))
    y = op.forward(dev(x), pair_mode=L.PAIR_GATE)
    assert rel_rms(y, ref) <= 2 * tol


def test_set_conv_math_on_modules(oracle):
    """modules.hipconv.set_conv_math: the generator in bf16 arithmetic stays within a bf16 tolerance of the default engine"""
    from visinger_amd.modules.hipconv import set_conv_math
    from visinger_amd.modules.visinger.decoder import Generator
    torch.manual_seed(0)
    g = Generator(80, "1", [3, 7], [[1, 3, 5], [1, 3, 5]], [4, 4], 64, [8, 8], gin_channels=0).cuda().eval()
    g.remove_weight_norm()
    x = torch.randn(2, 80, 50, device="cuda")
    with torch.no_grad():
        y6 = g(x)
        set_conv_math(g, L.MATH_F32)
        y0 = g(x)
        set_conv_math(g, L.MATH_BF16)
        y1 = g(x)
        set_conv_math(g, None)
    assert float((y6 - y0).abs().max()) <= 2e-5
    assert 1e-6 < float((y1 - y0).abs().max()) <= 5e-2


def _attention_fp64(qkv, nh, dk, rel_k, rel_v, mask, ws, share):
    """fp64 restatement of the attention core (rel_transformer.py:148-179; masked rows attend uniformly: -1e4 fill) on a fused q|k|v buffer."""
    B, C3, T = qkv.shape
    C = C3 // 3
    nrel = 0 if ws is None else 2 * ws + 1
    q, k, v = (qkv[:, i * C:(i + 1) * C].double().view(B, nh, dk, T).transpose(2, 3) for i in range(3))
    sc = (q / dk ** 0.5) @ k.transpose(-1, -2)
    idx = torch.arange(T)
    rel = idx[None, :] - idx[:, None]
    if nrel:
        rk = rel_k.double().expand(nh, nrel, dk) if share else rel_k.double()
        rv = rel_v.double().expand(nh, nrel, dk) if share else rel_v.double()
        qr = (q / dk ** 0.5) @ rk.transpose(-1, -2)[None]                        # [B, nh, T, nrel]
        band = rel.abs() <= ws
        sc = sc + torch.gather(qr, 3, (rel + ws).clamp(0, 2 * ws)[None, None].expand(B, nh, T, T)) * band
    am = mask.double()[:, None, :, None] * mask.double()[:, None, None, :]
    sc = sc.masked_fill(am == 0, -1e4)
    pr = torch.softmax(sc, -1)
    out = pr @ v
    if nrel:
        cols = idx[:, None] + torch.arange(-ws, ws + 1)[None, :]
        ok = (cols >= 0) & (cols < T)
        pw = torch.gather(pr, 3, cols.clamp(0, T - 1)[None, None].expand(B, nh, T, nrel)) * ok
        out = out + pw @ rv[None]
    return out.transpose(2, 3).reshape(B, C, T)


@pytest.mark.parametrize("math", [L.MATH_SPLIT6, L.MATH_SPLIT3])
@pytest.mark.parametrize("dk,nh,T,ws,share", [(96, 2, 1024, 4, True), (64, 2, 260, 4, False), (128, 1, 516, None, True), (32, 4, 64, 4, True),
                                              (96, 2, 36, 4, True), (80, 3, 132, 7, False)])
def test_split_attention_is_fp32_class(oracle, dk, nh, T, ws, share, math):
    """vs_relattn_fwd in the two split arithmetics of csrc/attention_bf16.hip -- VS_MATH_SPLIT6 (TERMS = 6: q / sqrt(dk), k, v and the
    probabilities split exactly into three bf16 planes, six cross products per product) and VS_MATH_SPLIT3 (TERMS = 3, the default
    arithmetic of the path since round 6: two f16 planes under power-of-two scales -- per query, per K tile, running per V tile -- three
    cross products) -- against the exact-fp32 MFMA kernel AND the fp64 restatement (rel_transformer.py:148-179): fp32-class agreement,
    the same bar the fp32 kernel is held to."""
    from visinger_amd.ops import rel_attention
    from visinger_amd import _lib
    g = torch.Generator().manual_seed(dk * 11 + T)
    B, C = 3, dk * nh
    qkv = torch.randn(B, 3 * C, T, generator=g)
    nrel = 0 if ws is None else 2 * ws + 1
    rel_k = (torch.randn(1 if share else nh, nrel, dk, generator=g) * dk ** -0.5) if nrel else None
    rel_v = (torch.randn(1 if share else nh, nrel, dk, generator=g) * dk ** -0.5) if nrel else None
    lens = torch.tensor([T, max(1, (2 * T) // 3), 0])
    mask = (torch.arange(T)[None] < lens[:, None]).float()
    cu = lambda t: None if t is None else t.cuda()
    ref32 = rel_attention(cu(qkv), nh, cu(rel_k), cu(rel_v), cu(mask), ws, math=L.MATH_F32)
    got = rel_attention(cu(qkv), nh, cu(rel_k), cu(rel_v), cu(mask), ws, math=math)
    inst = _lib.lib().vs_last_kernel_name().decode()
    assert inst.startswith("relattn_bf16_kernel<") and inst.endswith(", 6>" if math == L.MATH_SPLIT6 else ", 3>"), inst
    assert float((got - ref32).abs().max()) <= 2e-5
    ref64 = _attention_fp64(qkv, nh, dk, rel_k, rel_v, mask, ws, share)
    e6 = float((got.cpu().double() - ref64).abs().max())
    e32 = float((ref32.cpu().double() - ref64).abs().max())
    print(f"attention dk={dk} T={T} {inst}: max err vs fp64 split {e6:.2e}, fp32 MFMA {e32:.2e}")
    assert e6 <= 2e-5 and e6 <= 3.0 * e32 + 2e-6


@pytest.mark.parametrize("case", ["ramp_up", "ramp_down", "channel_gains", "outlier_keys", "tiny", "huge"])
def test_split_f16_attention_scales_follow_the_data(case):
    """The power-of-two scales of the split-f16 attention (csrc/attention_bf16.hip, TERMS = 3) on inputs that move them: K / V magnitudes that
    grow or shrink by 2^24 along the sequence (every K tile under its own scale; the V scale runs upwards and the output accumulators are
    rescaled when a tile raises it), per-channel gains over 2^-8 .. 2^8, single keys 2^12 above their neighbours, inputs near 2^-60 and
    near 2^40 (far outside the f16 range without the scales).  Bar: the fp32 MFMA kernel's own error against the fp64 restatement, three
    times over, relative to the largest output of the (batch, head)."""
    from visinger_amd.ops import rel_attention
    from visinger_amd import _lib
    dk, nh, T, ws, B = 96, 2, 516, 4, 2
    C = dk * nh
    g = torch.Generator().manual_seed(len(case))
    qkv = torch.randn(B, 3 * C, T, generator=g)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    t = torch.arange(T, dtype=torch.float32) / (T - 1)
    if case in ("ramp_up", "ramp_down"):
        e = 24.0 * (t if case == "ramp_up" else 1 - t) - 12.0
        v *= torch.exp2(e)[None, None]
        k *= torch.exp2(e / 8)[None, None]                       # (scores stay in a range where several keys share a row's weight)
        q *= 0.2
    elif case == "channel_gains":
        gain = torch.exp2(torch.rand(3 * C, generator=g) * 16 - 8)
        gain[:C] = 1.0 / gain[C:2 * C]                            # q against k: the products stay O(1), the operands do not
        qkv *= gain[None, :, None]
    elif case == "outlier_keys":
        pos = torch.randint(0, T, (12,), generator=g)
        v[:, :, pos] *= 4096.0
        k[:, :, pos] *= 4.0
    elif case == "tiny":
        v *= 2.0 ** -60
        k *= 2.0 ** -30
        q *= 2.0 ** 30
    elif case == "huge":
        v *= 2.0 ** 40
        k *= 2.0 ** 20
        q *= 2.0 ** -20
    rel_k = torch.randn(1, 2 * ws + 1, dk, generator=g) * dk ** -0.5
    rel_v = torch.randn(1, 2 * ws + 1, dk, generator=g) * dk ** -0.5 * float(v.abs().mean())
    mask = torch.ones(B, T)
    mask[1, (3 * T) // 4:] = 0
    cu = lambda x: x.cuda().contiguous()
    ref32 = rel_attention(cu(qkv), nh, cu(rel_k), cu(rel_v), cu(mask), ws, math=L.MATH_F32)
    got = rel_attention(cu(qkv), nh, cu(rel_k), cu(rel_v), cu(mask), ws, math=L.MATH_SPLIT3)
    assert _lib.lib().vs_last_kernel_name().decode() == "relattn_bf16_kernel<3, 32, 3>"
    assert bool(torch.isfinite(got).all())
    ref64 = _attention_fp64(qkv, nh, dk, rel_k, rel_v, mask, ws, True)
    scale = ref64.view(B, nh, dk, T).abs().amax(dim=(2, 3), keepdim=True).clamp_min(1e-300)
    rel = lambda x: float(((x.cpu().double() - ref64).view(B, nh, dk, T).abs() / scale).max())
    e3, e32 = rel(got), rel(ref32)
    print(f"split-f16 attention, {case}: max err / max|out| of the head: split-f16 {e3:.2e}, fp32 MFMA {e32:.2e}")
    assert e3 <= 3.0 * e32 + 1e-6


@pytest.mark.parametrize("kind,Cin,Cout,k,d_or_u,T,B", [(L.CONV1D, 256, 256, 7, 3, 1024, 2), (L.CONV1D, 128, 128, 11, 5, 700, 1), (L.CONV1D, 512, 256, 3, 1, 261, 2),
                                                       (L.CONV1D, 200, 136, 7, 1, 33, 3), (L.CONV1D, 64, 128, 3, 1, 4096, 1), (L.CONV1D, 40, 96, 5, 2, 1000, 1),
                                                       (L.CONV_TRANSPOSE1D, 512, 256, 16, 8, 96, 2), (L.CONV_TRANSPOSE1D, 128, 64, 4, 2, 515, 1),
                                                       (L.CONV1D, 192, 512, 7, 1, 40, 1), (L.CONV1D, 64, 64, 11, 3, 900, 2), (L.CONV1D, 32, 32, 7, 1, 2050, 1),
                                                       (L.CONV_TRANSPOSE1D, 64, 32, 4, 2, 600, 2)])
def test_bf16_resident_tensors_are_the_bf16_arithmetic_on_rounded_tensors(kind, Cin, Cout, k, d_or_u, T, B):
    """vs_dtype (include/visinger_hip.h): bf16-RESIDENT x / y / res / acc in the plain-bf16 arithmetic.  That arithmetic rounds every conv
    operand to bf16 while staging, so a launch on a bf16 tensor must equal -- BIT FOR BIT -- the launch on the same values held in fp32,
    and a bf16 output must be the round-to-nearest-even of the fp32 output (residual / accumulate / scale / ReLU applied in fp32 before the
    one rounding): the 128-row tile shape these tensors are supported on (>= 96 output rows), interior and ragged tiles (vector and element-wise epilogues), both sequence
    ends, transposed convs, input leaky-relu / mask, per-item bias."""
    from visinger_amd.ops import ConvOp
    g = torch.Generator().manual_seed(Cin + Cout * 3 + k + T)
    tr = kind == L.CONV_TRANSPOSE1D
    pad = (k - d_or_u) // 2 if tr else d_or_u * (k - 1) // 2
    op = ConvOp(kind, Cin, Cout, k, d_or_u, pad).set_math(L.MATH_BF16)
    w = torch.randn((Cin, Cout, k) if tr else (Cout, Cin, k), generator=g) / np.sqrt(Cin * k / (d_or_u if tr else 1))
    op.set_weights(w.cuda(), None, torch.randn(Cout, generator=g).cuda())
    xb = torch.randn(B, Cin, T, generator=g).cuda().bfloat16()
    Tout = op.out_len(T)
    mask = torch.ones(B, T)
    mask[-1, (2 * T) // 3:] = 0
    mask = mask.cuda()
    cond = torch.randn(B, Cout, generator=g).cuda()
    resb = torch.randn(B, Cout, Tout, generator=g).cuda().bfloat16()
    accb = torch.randn(B, Cout, Tout, generator=g).cuda().bfloat16()
    variants = [dict(in_act=L.IN_LRELU), dict(in_act=L.IN_LRELU_MASK, mask=mask, bias_b=cond)]
    if not tr:
        variants += [dict(in_act=L.IN_LRELU, res=True), dict(in_act=L.IN_LRELU, res=True, acc=True, scale=1.0 / 3, out_act=L.OUT_RELU),
                     dict(in_act=L.IN_LRELU_MASK, mask=mask, res=True, out_mask=True)]
    for kw in variants:
        kw = dict(kw)
        use_res, use_acc = kw.pop("res", False), kw.pop("acc", False)
        ref = op.forward(xb.float(), res=resb.float() if use_res else None, acc=accb.float() if use_acc else None, **kw)     # fp32 tensors
        # (short launches take smaller tiles; same sums, same order.  Round 5: a masked plain-bf16 launch takes a conv_ktap instance wherever one exists)
        assert op.kernel_instance().startswith(("conv_split_kernel<1, ", "conv_ktap_kernel<")), op.kernel_instance()
        wide = (d_or_u * Cout if tr else Cout) >= 96          # 128-row tiles: every combination; narrower: bf16 in AND out only
        if not use_res and not wide:
            with pytest.raises(L.VisingerHipError):
                op.forward(xb, **kw)
        if not use_res and wide:
            y1 = op.forward(xb, **kw)                                                                                        # bf16 in, fp32 out
            assert y1.dtype == torch.float32 and torch.equal(y1, ref), (kw, float((y1 - ref).abs().max()))
            assert op.kernel_instance() == "conv_split_kernel_bf16io<1, 8, 4, 1, 1, 1>", op.kernel_instance()
        y3 = op.forward(xb, res=resb if use_res else None, acc=accb if use_acc else None, y_dtype=torch.bfloat16, **kw)        # bf16 in / out
        assert y3.dtype == torch.bfloat16 and torch.equal(y3, ref.bfloat16()), (kw, float((y3.float() - ref).abs().max()))
        # (a transposed conv with bf16 in AND out runs the instance with the polyphase store path, csrc/conv_epilogue_tr_bf16.inc)
        # (round 5: the stride-1 convs of 3 / 7 / 11 taps behind a leaky-relu run conv_ktap_kernel<taps, transform, 1 plane, tensors 3, tile...>)
        ki = op.kernel_instance()
        assert (ki.startswith(("conv_split_kernel_bf16io<1, ", "conv_split_tr_kernel_bf16io<1, ")) and ki.endswith(", 1, 3>")) or \
            (ki.startswith("conv_ktap_kernel<") and ", 1, 3, 4, 1, 8, 1>" in ki), ki
        if not use_res and wide:
            y2 = op.forward(xb.float(), y_dtype=torch.bfloat16, **kw)                                                        # fp32 in, bf16 out
            assert torch.equal(y2, ref.bfloat16()) and op.kernel_instance() == "conv_split_kernel_bf16io<1, 8, 4, 1, 1, 2>"
    # loud refusals: another arithmetic, mismatched residual type (and above: fp32 <-> bf16 conversions on tiles of fewer than 96 rows)
    op6 = ConvOp(kind, Cin, Cout, k, d_or_u, pad).set_math(L.MATH_SPLIT6)
    op6.set_weights(w.cuda(), None, None)
    with pytest.raises(L.VisingerHipError):
        op6.forward(xb)
    if not tr:
        with pytest.raises(L.VisingerHipError):
            op.forward(xb, res=resb.float(), y_dtype=torch.bfloat16)


@pytest.mark.parametrize("C,k,d,T,B", [(32, 3, 1, 4096, 2), (32, 7, 3, 3000, 1), (32, 11, 5, 1030, 2), (64, 3, 3, 2048, 1), (64, 7, 1, 777, 2), (64, 11, 5, 4100, 1)])
def test_fused_pair_and_conv_post_on_bf16_resident_tensors(C, k, d, T, B):
    """The fused residual pair (decoder.py:92-101) and the 32 -> 1 output conv (decoder.py:55-57) on bf16-RESIDENT tensors in the plain-bf16
    arithmetic: bit for bit the launch on the same values held in fp32, the pair's output rounded to nearest even once (x staged from
    bf16, residual and MRF accumulator read as bf16, interior and ragged tiles, vector and element-wise epilogues)."""
    from visinger_amd.ops import ConvOp, respair_forward, respair_supported
    g = torch.Generator().manual_seed(C * 7 + k + d + T)
    ops_ = []
    for dd in (d, 1):
        op = ConvOp(L.CONV1D, C, C, k, dd, dd * (k - 1) // 2).set_math(L.MATH_BF16)
        op.set_weights((torch.randn(C, C, k, generator=g) / np.sqrt(C * k)).cuda(), None, torch.randn(C, generator=g).cuda())
        ops_.append(op)
    assert respair_supported(ops_[0], ops_[1])
    xb = torch.randn(B, C, T, generator=g).cuda().bfloat16()
    accb = torch.randn(B, C, T, generator=g).cuda().bfloat16()
    for use_acc, scale in ((False, 1.0), (True, 1.0 / 3)):
        ref = respair_forward(ops_[0], ops_[1], xb.float(), torch.empty(B, C, T, device="cuda"), res=xb.float(),
                              acc=accb.float() if use_acc else None, scale=scale)
        assert ops_[0].kernel_instance() in ("respair_split_kernel<2, 1, 4, 1, false>", "respair_split_kernel<2, 2, 2, 1, false>")
        y = respair_forward(ops_[0], ops_[1], xb, torch.empty_like(xb), res=xb, acc=accb if use_acc else None, scale=scale)
        assert ops_[0].kernel_instance().endswith(", 1, true>"), ops_[0].kernel_instance()
        assert y.dtype == torch.bfloat16 and torch.equal(y, ref.bfloat16()), float((y.float() - ref).abs().max())
    with pytest.raises(L.VisingerHipError):
        respair_forward(ops_[0], ops_[1], xb, torch.empty(B, C, T, device="cuda"), res=xb)          # y fp32, x bf16
    if C == 32:
        post = ConvOp(L.CONV1D, 32, 1, 7, 1, 3).set_math(L.MATH_BF16)
        post.set_weights((torch.randn(1, 32, 7, generator=g) / 15).cuda(), None, None)
        w32 = post.forward(xb.float(), in_act=L.IN_LRELU, out_act=L.OUT_TANH)
        wb = post.forward(xb, in_act=L.IN_LRELU, out_act=L.OUT_TANH)
        assert wb.dtype == torch.float32 and post.kernel_instance().startswith("conv_small_kernel") and torch.equal(wb, w32)


@pytest.mark.parametrize("dk,nh,T,ws,share,B,math", [(96, 2, 1024, 4, True, 1, L.MATH_SPLIT6), (96, 2, 516, 4, True, 2, L.MATH_SPLIT6), (64, 2, 260, 4, False, 1, L.MATH_SPLIT6),
                                                     (96, 2, 1024, 4, True, 1, L.MATH_SPLIT3), (96, 2, 516, 4, True, 2, L.MATH_SPLIT3), (64, 2, 260, 4, False, 1, L.MATH_SPLIT3),
                                                     (128, 1, 2048, None, True, 1, L.MATH_SPLIT3),
                                                     (128, 1, 2048, None, True, 1, L.MATH_SPLIT6), (96, 2, 1024, 4, True, 1, L.MATH_BF16),
                                                     (256, 2, 1024, 4, True, 1, L.MATH_BF16)])
def test_key_split_attention_equals_one_pass(dk, nh, T, ws, share, B, math):
    """vs_relattn_fwd_ksplit: the keys of a (batch, head) cut into ranges that run as separate workgroups and are merged by a second kernel
    (single utterances: B * heads * T / 128 workgroups do not fill the chip) -- same scores, same relative terms, same -1e4 fill; the
    merged rows equal the one-pass rows up to the reassociation of the row sums (ragged mask with the whole tail of the keys padded,
    key-tile counts that do not divide by the number of ranges, shared and per-head tables, no window)."""
    from visinger_amd.ops import rel_attention
    g = torch.Generator().manual_seed(dk + T + nh)
    C = nh * dk
    qkv = torch.randn(B, 3 * C, T, generator=g).cuda()
    R = 2 * ws + 1 if ws is not None else 0
    rel_k = (torch.randn(1 if share else nh, R, dk, generator=g) * dk ** -0.5).cuda() if ws is not None else None
    rel_v = (torch.randn(1 if share else nh, R, dk, generator=g) * dk ** -0.5).cuda() if ws is not None else None
    mask = torch.ones(B, T)
    mask[-1, (3 * T) // 5:] = 0
    mask = mask.cuda()
    one = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=math, ksplit_auto=False)
    inst = L.lib().vs_last_kernel_name().decode()
    split = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=math, ksplit_auto=True)
    assert inst.startswith(("relattn_bf16_kernel<", "relattn_dma_kernel<")) and bool(torch.isfinite(split).all())
    tol = 2e-6 if math in (L.MATH_SPLIT6, L.MATH_SPLIT3) else 2e-3
    assert float((split - one).abs().max()) <= tol * max(1.0, float(one.abs().max())), float((split - one).abs().max())


@pytest.mark.parametrize("dk,nh,T,ws,share,B", [(256, 2, 1028, 4, True, 2), (96, 2, 1024, 4, True, 3), (64, 2, 1100, 4, False, 2), (128, 1, 2048, None, True, 1),
                                                (192, 1, 1032, 7, True, 2), (32, 4, 1024, 4, True, 1), (256, 2, 4096, 4, True, 1)])
def test_prepacked_kv_attention_is_bit_identical(vs_option, dk, nh, T, ws, share, B):
    """vs_relattn_fwd_work (ABI 5): on sequences of 1024 frames and more the plain-bf16 attention kernel takes its K / V tiles as LDS images
    packed once per launch (attn_pack_kv_kernel) instead of converting fp32 -> bf16 in every query block.  The same bf16 operands in the
    same order: the output equals the in-place kernel's bit for bit -- every head-width instance, key tiles cut by T, ragged masks with an
    all-padding item, with and without the key split; below 1024 frames and with VS_NO_ATTN_KVPACK the library asks for no scratch."""
    vs_option("VS_NO_ATTN_DMA", 1)      # (this test is about relattn_bf16_kernel's packed path; the LDS-DMA kernel of round 4 sums the score tile in another order: tests/test_attention_dma_gpu.py)
    from visinger_amd.ops import rel_attention
    g = torch.Generator().manual_seed(dk * 3 + T)
    C = dk * nh
    qkv = torch.randn(B, 3 * C, T, generator=g).cuda()
    nrel = 0 if ws is None else 2 * ws + 1
    rel_k = (torch.randn(1 if share else nh, nrel, dk, generator=g) * dk ** -0.5).cuda() if nrel else None
    rel_v = (torch.randn(1 if share else nh, nrel, dk, generator=g) * dk ** -0.5).cuda() if nrel else None
    lens = torch.tensor([T, (2 * T) // 3, 0][:B])
    mask = (torch.arange(T)[None] < lens[:, None]).float().cuda()
    lib = L.lib()
    assert lib.vs_relattn_kv_work_bytes(B, nh, dk, T, L.MATH_BF16) > 0
    assert lib.vs_relattn_kv_work_bytes(B, nh, dk, 512, L.MATH_BF16) == 0 and lib.vs_relattn_kv_work_bytes(B, nh, dk, T, L.MATH_SPLIT6) == 0
    for auto in (False, True):
        packed = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=auto)
        name = lib.vs_last_kernel_name().decode()
        assert name.startswith("relattn_bf16_kernel<") and name.endswith(", true>"), name
        vs_option("VS_NO_ATTN_KVPACK", 1)
        assert lib.vs_relattn_kv_work_bytes(B, nh, dk, T, L.MATH_BF16) == 0
        plain = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=auto)
        assert not lib.vs_last_kernel_name().decode().endswith(", true>")
        vs_option("VS_NO_ATTN_KVPACK", 0)
        assert torch.equal(packed, plain)


@pytest.mark.parametrize("cin,cout,k,u,T,B", [(128, 64, 4, 2, 8192, 8), (64, 32, 4, 2, 30000, 4), (256, 128, 16, 8, 1024, 8), (512, 256, 16, 8, 300, 4),
                                              (128, 64, 7, 3, 7777, 4), (96, 32, 11, 5, 5150, 6)])
def test_transposed_conv_polyphase_store_path_is_bit_identical(vs_option, cin, cout, k, u, T, B):
    """conv_split_tr_kernel (csrc/conv_epilogue_tr.inc): the transposed convs between the generator stages store their polyphase outputs
    with one buffer_store per element (per-lane column offset + wave-uniform row offset) on interior tiles and fall through to the generic
    epilogue at the edges.  Same accumulators, same finishing fma: bit-identical to the generic instance (VS_NO_TR_EPI), and within the
    arithmetic's error of the fp64 oracle -- the generator's own strides (2, 8), the hop-300 generator's (3, 5), lengths that end inside
    a tile, item counts > 1."""
    from visinger_amd.ops import ConvOp
    g = torch.Generator().manual_seed(cin + k + T)
    pad = (k - u) // 2
    op = ConvOp(L.CONV_TRANSPOSE1D, cin, cout, k, u, pad)
    w = (torch.randn(cin, cout, k, generator=g) * (cin * k / u) ** -0.5).cuda()
    bias = (0.1 * torch.randn(cout, generator=g)).cuda()
    op.set_weights(w, None, bias)
    x = torch.randn(B, cin, T, generator=g).cuda()
    y = op.forward(x, in_act=L.IN_LRELU)
    name = op.kernel_instance()
    assert name.startswith(("conv_split_tr_kernel<", "conv_ktap_kernel<2, 1, 2, 4, ")), name      # (k = 2 * stride on a chip-filling grid: the conv_ktap instance, same epilogue)
    vs_option("VS_NO_TR_EPI", 1)
    y0 = op.forward(x, in_act=L.IN_LRELU)
    assert op.kernel_instance().startswith("conv_split_kernel<")
    assert torch.equal(y, y0)
    ref = torch.nn.functional.conv_transpose1d(torch.nn.functional.leaky_relu(x.double(), 0.1), w.double(), bias.double(), stride=u, padding=pad)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    # the same on bf16-RESIDENT tensors in the plain-bf16 arithmetic (BASELINE config 5): conv_split_tr_kernel_bf16io against its generic instance
    vs_option("VS_NO_TR_EPI", 0)
    op.set_math(L.MATH_BF16)
    xb = x.bfloat16()
    yb = op.forward(xb, in_act=L.IN_LRELU, y_dtype=torch.bfloat16)
    if u * cout >= 64:
        assert op.kernel_instance().startswith("conv_split_tr_kernel_bf16io<"), op.kernel_instance()
    vs_option("VS_NO_TR_EPI", 1)
    yb0 = op.forward(xb, in_act=L.IN_LRELU, y_dtype=torch.bfloat16)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb, yb0)
    assert float((yb.double() - ref).abs().max()) <= 3e-2 * float(ref.abs().max())


def test_prepacked_kv_attention_replays_from_a_hip_graph():
    """ops.rel_attention takes the scratch of the K / V tile images from torch's caching allocator, so the packed path stays capturable:
    pack kernel + attention kernel recorded into a HIP graph, replayed on NEW q | k | v values in the static input buffer, bit-identical
    to an eager launch on those values."""
    from visinger_amd.ops import rel_attention
    g = torch.Generator().manual_seed(11)
    B, nh, dk, T, ws = 2, 2, 64, 1024, 4
    C = nh * dk
    qkv = torch.randn(B, 3 * C, T, generator=g).cuda()
    rel_k = (torch.randn(1, 2 * ws + 1, dk, generator=g) * dk ** -0.5).cuda()
    rel_v = (torch.randn(1, 2 * ws + 1, dk, generator=g) * dk ** -0.5).cuda()
    mask = (torch.arange(T)[None] < torch.tensor([T, 700])[:, None]).float().cuda()
    out = torch.empty(B, C, T, device="cuda")

    def step():
        return rel_attention(qkv, nh, rel_k, rel_v, mask, ws, out=out, math=L.MATH_BF16, ksplit_auto=False)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    assert L.lib().vs_last_kernel_name().decode().endswith(", true>")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    new = torch.randn(B, 3 * C, T, generator=g).cuda()
    qkv.copy_(new)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    replayed = out.clone()
    assert torch.equal(replayed, rel_attention(new, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=False))


def test_bf16_resident_tensors_random_sweep():
    """Randomised version of the two tests above: 160 random (shape, dilation / stride, fused option) cases of the plain-bf16 conv on
    bf16-RESIDENT tensors, each bit for bit against the same launch on the same values held in fp32 (output rounded once)."""
    from visinger_amd.ops import ConvOp
    r = np.random.default_rng(2024)
    g = torch.Generator().manual_seed(2024)
    done = 0
    for case in range(160):
        tr = r.random() < 0.25
        B = int(r.integers(1, 4))
        if tr:
            u = int(r.choice([2, 3, 5, 8]))
            k = int(u * r.integers(1, 3) + r.integers(0, 2))
            if (k - u) % 2:
                k += 1
            Cin = int(r.integers(8, 300))
            Cout = int(r.choice([32, 64, 128, 256])) if r.random() < 0.7 else int(r.integers(20, 200))
            T = int(r.integers(1, 400))
            op = ConvOp(L.CONV_TRANSPOSE1D, Cin, Cout, k, u, (k - u) // 2).set_math(L.MATH_BF16)
            w = torch.randn(Cin, Cout, k, generator=g) / np.sqrt(Cin * k / u)
            rows = u * Cout
        else:
            k = int(r.choice([1, 3, 5, 7, 9, 11]))
            d = int(r.choice([1, 2, 3, 5]))
            Cin, Cout = int(r.integers(4, 300)), int(r.choice([32, 64, 128, 192, 256, 384])) if r.random() < 0.7 else int(r.integers(5, 300))
            T = int(r.integers(1, 3000))
            op = ConvOp(L.CONV1D, Cin, Cout, k, d, d * (k - 1) // 2).set_math(L.MATH_BF16)
            w = torch.randn(Cout, Cin, k, generator=g) / np.sqrt(Cin * k)
            rows = Cout
        op.set_weights(w.cuda(), None, torch.randn(Cout, generator=g).cuda() if r.random() < 0.8 else None)
        xb = torch.randn(B, Cin, T, generator=g).cuda().bfloat16()
        Tout = op.out_len(T)
        kw = dict(in_act=int(r.choice([L.IN_NONE, L.IN_LRELU])))
        mask = None
        if r.random() < 0.4:
            mask = (torch.rand(B, T, generator=g) < 0.8).float().cuda()
            kw.update(in_act=L.IN_LRELU_MASK if kw["in_act"] == L.IN_LRELU else L.IN_MASK, mask=mask)
        res = acc = None
        if not tr:
            if r.random() < 0.6:
                res = torch.randn(B, Cout, Tout, generator=g).cuda().bfloat16()
            if r.random() < 0.3:
                acc = torch.randn(B, Cout, Tout, generator=g).cuda().bfloat16()
                kw.update(scale=float(r.choice([1.0, 1.0 / 3])))
            if r.random() < 0.3:
                kw.update(out_act=L.OUT_RELU)
            if mask is not None and r.random() < 0.5:
                kw.update(out_mask=True)
        ref = op.forward(xb.float(), res=None if res is None else res.float(), acc=None if acc is None else acc.float(), **kw)
        wide = rows >= 96                                            # 128-row tiles: every combination of element types
        try:
            y = op.forward(xb, res=res, acc=acc, y_dtype=torch.bfloat16, **kw)
        except L.VisingerHipError:
            assert rows < 32 or (rows % 128 == 64 and rows > 128) or rows <= 4, (case, rows)      # tile shapes without a bf16 instance: loud
            continue
        assert torch.equal(y, ref.bfloat16()), (case, tr, Cin, Cout, k, T, kw.keys(), float((y.float() - ref).abs().max()))
        done += 1
        if wide and res is None and acc is None:
            try:
                y1 = op.forward(xb, **kw)
            except L.VisingerHipError:
                continue
            assert torch.equal(y1, ref), (case, "bf16 in / fp32 out")
    assert done >= 100, done
