"""conv_ktap_kernel (csrc/conv_ktap.hip, round 5): the 128 x 256 tile of the split-f16 x3 conv engine with the taps unrolled and the staging of the
next 16-channel chunk dealt out over the MFMA gaps of the current one.  Its outputs must be BIT-IDENTICAL to conv_split_kernel<1, 8, 4, 1, 3> (same
operand planes, same running tile scale, same MFMA order, same epilogue) -- the instance every parity statement of rounds 3-4 about the wide generator
convs was made on (reference work: modules/visinger/decoder.py:72-101, modules/rel_transformer.py:336-345) -- and stay within the engine's fp32-class bound
against an fp64 convolution."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(op, x, res, acc, scale, in_act, out_act, mask=None, out_mask=False, bias_b=None):
    y = torch.empty((x.shape[0], op.rows_out, x.shape[2]), device=x.device)
    op.forward(x, y=y, res=res, acc=acc, scale=scale, in_act=in_act, out_act=out_act, mask=mask, out_mask=out_mask, bias_b=bias_b)
    return y, op.kernel_instance()


CASES = [
    # C_in, C_out, k, dil, B, T, res, acc, scale, in_act (0 none, 1 lrelu, 2 mask, 3 lrelu + mask), out_act, out_mask, bias_b
    (128, 128, 7, 1, 4, 32768, True, False, 1.0, 1, 0, False, False),          # ResBlock1 conv2 (decoder.py:100-103), 128 channels
    (128, 128, 7, 5, 4, 32768, False, False, 1.0, 1, 0, False, False),         # conv1, dilation 5
    (128, 128, 11, 3, 5, 26368, True, True, 1.0 / 3.0, 1, 0, False, False),    # last conv of a block: + MRF accumulator, * 1/3
    (128, 128, 11, 5, 2, 8192 + 100, True, False, 1.0, 1, 0, False, False),    # ragged last tile (element-wise epilogue), widest window (span 50)
    (128, 128, 3, 1, 3, 44032, True, False, 1.0, 0, 0, False, False),          # k = 3: the densest schedule (55 micro-operations in 72 gaps)
    (256, 256, 7, 3, 2, 32768, True, False, 1.0, 1, 0, False, False),          # 256 channels: two row blocks, 16 chunks
    (256, 256, 3, 5, 9, 8192, True, True, 0.5, 1, 2, False, False),            # short items: every other tile an edge tile; relu
    (192, 512, 7, 1, 6, 1024, False, False, 1.0, 2, 0, False, True),           # conv_pre: masked input, per-item conditioning bias (decoder.py:41-43), odd chunk count 12 -> even, T = 4 tiles
    (208, 128, 7, 1, 3, 4096, False, False, 1.0, 3, 1, True, False),           # 13 chunks (odd: the loop leaves after its first half), lrelu + mask, tanh, output mask
    (128, 768, 11, 1, 4, 2048, False, False, 1.0, 2, 2, True, False),          # FFN conv_1 shape family: masked in and out, relu
    (192, 768, 9, 1, 8, 1024, False, False, 1.0, 2, 2, True, False),           # FFN conv_1 itself (rel_transformer.py:336-345): k = 9
    (144, 100, 3, 2, 2, 5000, True, False, 1.0, 1, 0, False, False),           # 100 output rows (padded row tile), T not a multiple of 4 columns x 256
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%d-%d_k%d_d%d_B%d_T%d_a%d%s%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[9], "_res" if c[6] else "", "_acc" if c[7] else ""))
def test_ktap_kernel_is_bit_identical_to_the_tile_kernel(case, vs_option):
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    cin, cout, k, d, B, T, use_res, use_acc, scale, in_act, out_act, out_mask, use_bb = case
    vs_option("VS_CONV_MATH", 3)
    vs_option("VS_NO_SMALL_GRID", 1)                                             # (the 128-row tile whatever the grid size)
    g = torch.Generator(device="cuda").manual_seed(1000 + k * 10 + d)
    op = ConvOp(L.CONV1D, cin, cout, k, d, (k * d - d) // 2)
    w = torch.randn(cout, cin, k, device="cuda", generator=g) * (cin * k) ** -0.5
    bias = torch.randn(cout, device="cuda", generator=g) * 0.1
    op.set_weights(w, None, bias)
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    x[:, : cin // 2] *= torch.exp2(torch.randint(-6, 7, (B, cin // 2, 1), device="cuda", generator=g).float())      # per-channel scales: the running tile exponent moves
    x[0, 3, 1000:1300] = 2.0 ** 9                                                                                       # ... and rescales accumulators mid-tile
    x[0, cin - 2, 300:340] = -(2.0 ** 12)                                                                               # (a late chunk raises it again; negative: lrelu shrinks it)
    res = torch.randn(B, cout, T, device="cuda", generator=g) if use_res else None
    acc = torch.randn(B, cout, T, device="cuda", generator=g) if use_acc else None
    bias_b = torch.randn(B, cout, device="cuda", generator=g) if use_bb else None
    mask = None
    if in_act >= 2 or out_mask:
        lens = torch.randint(T // 2, T + 1, (B,), device="cuda", generator=g)
        lens[0] = T
        mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).float()
    ia = (L.IN_NONE, L.IN_LRELU, L.IN_MASK, L.IN_LRELU_MASK)[in_act]
    vs_option("VS_NO_KTAP", 1)
    y_ref, k_ref = _run(op, x, res, acc, scale, ia, out_act, mask, out_mask, bias_b)
    vs_option("VS_NO_KTAP", 0)
    y_new, k_new = _run(op, x, res, acc, scale, ia, out_act, mask, out_mask, bias_b)
    assert k_ref == "conv_split_kernel<1, 8, 4, 1, 3>" and k_new == "conv_ktap_kernel<%d, %d, 2, 0, 4, 1, 8, 1>" % (k, in_act), (k_ref, k_new)
    assert torch.equal(y_new, y_ref), float((y_new - y_ref).abs().max())
    y2, _ = _run(op, x, res, acc, scale, ia, out_act, mask, out_mask, bias_b)                # and run-to-run
    assert torch.equal(y2, y_new)
    # against fp64 on the first columns of item 0 (left edge tile included): the engine's fp32-class bound (tests/test_conv_split_gpu.py)
    n = min(4096, T)
    xs = x[:1, :, :min(T, 2 * n)].double()
    if mask is not None and in_act >= 2:
        xs = xs * mask[:1, None, :xs.shape[2]].double()
    xs = torch.where(xs > 0, xs, 0.1 * xs) if in_act in (1, 3) else xs
    ref = torch.nn.functional.conv1d(xs, w.double(), bias.double(), padding=(k * d - d) // 2, dilation=d)[:, :, :n]
    if use_bb:
        ref = ref + bias_b[:1, :, None].double()
    if use_res:
        ref = ref + res[:1, :, :n].double()
    if use_acc:
        ref = ref + acc[:1, :, :n].double()
    ref = ref * scale
    ref = torch.tanh(ref) if out_act == 1 else (torch.relu(ref) if out_act == 2 else ref)
    if out_mask:
        ref = ref * mask[:1, None, :n].double()
    err = (y_new[:1, :, :n].double() - ref)
    assert float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) <= 2e-6, float(err.abs().max())


BF16_CASES = [
    # C_in, C_out, k, dil, B, T, res, acc, scale, in_act, out_act, out_mask, bias_b, x bf16, y bf16
    (512, 1536, 1, 1, 2, 4096, False, False, 1.0, 2, 0, False, False, False, False),     # fused q | k | v projection at hidden 512 (rel_transformer.py:120-122), masked input
    (512, 512, 1, 1, 3, 2048 + 64, True, False, 1.0, 0, 0, True, False, False, False),   # conv_o + ragged last tile + output mask + residual
    (512, 2048, 9, 1, 2, 4096, False, False, 1.0, 2, 2, True, False, False, False),      # FFN conv_1 (rel_transformer.py:336-345): k 9, relu, masks
    (2048, 512, 9, 1, 2, 2048, False, False, 1.0, 2, 0, True, False, False, False),      # FFN conv_2: 128 chunks
    (256, 256, 7, 3, 2, 16384, True, False, 1.0, 1, 0, False, False, True, True),        # generator ResBlock1 conv on bf16-RESIDENT tensors (decoder.py:91-104)
    (128, 128, 11, 5, 3, 32768, True, True, 1.0 / 3.0, 1, 0, False, False, True, True),  # last conv of a block: + MRF accumulator, * 1/3, all bf16-resident
    (128, 128, 3, 1, 2, 8192 + 40, False, False, 1.0, 1, 0, False, False, True, True),   # ragged tail, bf16-resident
    (128, 128, 7, 1, 2, 8192, True, False, 1.0, 1, 0, False, False, False, False),       # bf16 arithmetic on fp32 tensors
    (192, 512, 7, 1, 3, 1024, False, False, 1.0, 2, 0, False, True, False, False),       # conv_pre with its conditioning bias, fp32 out
]


@pytest.mark.parametrize("case", BF16_CASES, ids=lambda c: "c%d-%d_k%d_d%d_B%d_T%d_a%d_io%d%d" % (c[0], c[1], c[2], c[3], c[4], c[5], c[9], c[13], c[14]))
def test_ktap_bf16_instances_are_bit_identical_to_the_tile_kernels(case, vs_option):
    """the plain-bf16 instances (csrc/conv_ktap_bf16.hip: BASELINE configs[4]) against conv_split_kernel<1, 8, 4, 1, 1> / conv_split_kernel_bf16io<1, 8, 4, 1, 1, 3>
    bit for bit, and against fp64 within the bf16 arithmetic's stated bound (rms <= 1e-2 of the output rms).
    Masked input transforms: in round 5 the tile kernel's plain-bf16 instances gave run-to-run different results there (a packed multiply with op_sel on its second
    source, wrong in lanes 48-63 beside MFMAs on gfx950: root-caused and removed in round 6, tests/test_conv_mask_race_gpu.py); both kernels are now held to each
    other bit for bit, and the new one to fp64, to itself run-to-run and to its own unmasked result on the items whose mask is all ones."""
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    cin, cout, k, d, B, T, use_res, use_acc, scale, in_act, out_act, out_mask, use_bb, xb, yb = case
    vs_option("VS_CONV_MATH", 1)
    vs_option("VS_NO_SMALL_GRID", 1)
    g = torch.Generator(device="cuda").manual_seed(2000 + k * 10 + d)
    op = ConvOp(L.CONV1D, cin, cout, k, d, (k * d - d) // 2)
    w = torch.randn(cout, cin, k, device="cuda", generator=g) * (cin * k) ** -0.5
    bias = torch.randn(cout, device="cuda", generator=g) * 0.1
    op.set_weights(w, None, bias)
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    res = torch.randn(B, cout, T, device="cuda", generator=g) if use_res else None
    acc = torch.randn(B, cout, T, device="cuda", generator=g) if use_acc else None
    if xb:
        x = x.bfloat16()
    if yb:
        res = None if res is None else res.bfloat16()
        acc = None if acc is None else acc.bfloat16()
    bias_b = torch.randn(B, cout, device="cuda", generator=g) if use_bb else None
    mask = None
    if in_act >= 2 or out_mask:
        lens = torch.randint(T // 2, T + 1, (B,), device="cuda", generator=g)
        lens[0] = T
        mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).float()
    ia = (L.IN_NONE, L.IN_LRELU, L.IN_MASK, L.IN_LRELU_MASK)[in_act]

    def run():
        y = torch.empty((B, cout, T), device="cuda", dtype=torch.bfloat16 if yb else torch.float32)
        op.forward(x, y=y, res=res, acc=acc, scale=scale, in_act=ia, out_act=out_act, mask=mask, out_mask=out_mask, bias_b=bias_b)
        return y, op.kernel_instance()

    vs_option("VS_NO_KTAP", 1)
    y_ref, k_ref = run()
    vs_option("VS_NO_KTAP", 0)
    y_new, k_new = run()
    io = (1 if xb else 0) | (2 if yb else 0)
    assert k_ref == ("conv_split_kernel_bf16io<1, 8, 4, 1, 1, %d>" % io if io else "conv_split_kernel<1, 8, 4, 1, 1>"), k_ref
    assert k_new == "conv_ktap_kernel<%d, %d, 1, %d, 4, 1, 8, 1>" % (k, in_act, io), k_new
    assert torch.equal(y_new, y_ref), float((y_new.float() - y_ref.float()).abs().max())      # (masked inputs included since round 6: see below)
    if in_act >= 2:           # item 0's mask is all ones: masking it is the identity
        y_plain = torch.empty_like(y_new[:1])
        op.forward(x[:1], y=y_plain, res=None if res is None else res[:1], acc=None if acc is None else acc[:1], scale=scale, in_act=ia - 2, out_act=out_act,
                   mask=mask[:1] if out_mask else None, out_mask=out_mask, bias_b=None if bias_b is None else bias_b[:1])
        assert op.kernel_instance().startswith("conv_ktap_kernel<") or op.kernel_instance().startswith("conv_split_kernel"), op.kernel_instance()
        assert torch.equal(y_plain, y_new[:1]), float((y_plain.float() - y_new[:1].float()).abs().max())
    y2, _ = run()
    assert torch.equal(y2, y_new)
    n = min(2048, T)
    xs = x[:1, :, :min(T, 2 * n)].double()
    if mask is not None and in_act >= 2:
        xs = xs * mask[:1, None, :xs.shape[2]].double()
    xs = torch.where(xs > 0, xs, 0.1 * xs) if in_act in (1, 3) else xs
    ref = torch.nn.functional.conv1d(xs, w.double(), bias.double(), padding=(k * d - d) // 2, dilation=d)[:, :, :n]
    if use_bb:
        ref = ref + bias_b[:1, :, None].double()
    if use_res:
        ref = ref + res[:1, :, :n].double()
    if use_acc:
        ref = ref + acc[:1, :, :n].double()
    ref = ref * scale
    ref = torch.tanh(ref) if out_act == 1 else (torch.relu(ref) if out_act == 2 else ref)
    if out_mask:
        ref = ref * mask[:1, None, :n].double()
    err = (y_new[:1, :, :n].double() - ref)
    assert float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) <= 1e-2, float(err.abs().max())


SMALL_CASES = [
    # C_in, C_out, k, dil, B, T, res, in_act, out_act, out_mask, flags (4 = ADJOINT: a grad-input conv), split3 (else bf16)
    (192, 192, 1, 1, 16, 512, False, 0, 0, False, 0, True),        # training step: 1 x 1 projections (32 x 128 tiles: 64 column tiles x 6 row tiles)
    (384, 192, 5, 1, 16, 512, False, 0, 0, False, 4, True),        # grad-input of the WaveNet's k = 5 in_layers (modules/visinger/encoder.py:158-161)
    (768, 192, 9, 1, 16, 512, False, 0, 0, False, 4, True),        # grad-input of FFN conv_1
    (192, 768, 9, 1, 16, 512, False, 0, 0, False, 0, True),        # FFN conv_1 forward (64 x 256 tiles)
    (1024, 1024, 5, 1, 1, 2468, False, 0, 0, False, 0, True),      # MultiPeriodDiscriminator layer, phase-stacked (modules/discriminator.py:28-47), one folded item
    (1536, 1024, 2, 1, 1, 1937, False, 0, 0, False, 0, True),      # ... its stride-3 layers as 2-tap convs over 3 x 512 stacked channels
    (256, 256, 11, 1, 16, 256, True, 0, 0, False, 0, True),        # generator resblock conv on a 32-frame segment, + residual
    (128, 128, 7, 3, 16, 2048, True, 0, 0, False, 4, True),        # its grad-input at the next stage, dilation 3
    (768, 192, 1, 1, 32, 1024, False, 2, 0, True, 0, True),        # inference: FFN conv_2 with masks (64 x 256)
    (192, 768, 9, 1, 32, 128, False, 2, 2, True, 0, True),         # text encoder's FFN conv_1 at T_ph = 128, relu
    (192, 576, 1, 1, 32, 1024, False, 0, 0, False, 0, True),       # fused q | k | v
    (176, 100, 3, 1, 3, 700, True, 0, 1, False, 0, True),          # padded row tile, ragged columns, tanh
    (2048, 512, 1, 1, 8, 512, False, 2, 0, True, 0, False),        # plain bf16 (BASELINE configs[4]): text encoder at hidden 512, masked
    (512, 2048, 9, 1, 2, 512, False, 2, 2, True, 0, False),        # ... its FFN conv_1
    (512, 512, 1, 1, 1, 4096, True, 0, 0, False, 0, False),        # single item: 32 x 128 tiles
]


@pytest.mark.parametrize("case", SMALL_CASES, ids=lambda c: "c%d-%d_k%d_d%d_B%d_T%d_a%d_f%d_%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[7], c[10], "s3" if c[11] else "bf"))
def test_ktap_small_tiles_are_bit_identical_to_the_tile_kernels(case, vs_option):
    """the 64 x 256 and 32 x 128 instances (csrc/conv_ktap_small.hip, conv_ktap_bf16.hip) that short launches dispatch to -- T_mel-sized tensors, single items, the
    training step's forward and grad-input convs -- against conv_split_kernel<1, 4, 2, 2, *> / <1, 1, 1, 4, *> bit for bit (split-f16; plain bf16 unmasked), and
    against fp64"""
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    cin, cout, k, d, B, T, use_res, in_act, out_act, out_mask, flags, s3 = case
    vs_option("VS_CONV_MATH", 3 if s3 else 1)
    g = torch.Generator(device="cuda").manual_seed(3000 + k * 10 + d + cin)
    adj = flags == 4
    op = ConvOp(L.CONV1D, cin, cout, k, d, (k * d - d) // 2, flags)
    # (an ADJOINT handle is handed the FORWARD conv's weight [C_in_fwd = cout, ...]: its own conv is the grad-input of that one)
    w = torch.randn(*((cin, cout, k) if adj else (cout, cin, k)), device="cuda", generator=g) * (cin * k) ** -0.5
    bias = None if adj else torch.randn(cout, device="cuda", generator=g) * 0.1
    op.set_weights(w, None, bias)
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    x[:, : cin // 2] *= torch.exp2(torch.randint(-6, 7, (B, cin // 2, 1), device="cuda", generator=g).float())
    x[0, cin - 2, T // 3: T // 3 + 40] = -(2.0 ** 11)
    res = torch.randn(B, cout, T, device="cuda", generator=g) if use_res else None
    mask = None
    if in_act >= 2 or out_mask:
        lens = torch.randint(T // 2, T + 1, (B,), device="cuda", generator=g)
        lens[0] = T
        mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).float()
    ia = (L.IN_NONE, L.IN_LRELU, L.IN_MASK, L.IN_LRELU_MASK)[in_act]

    def run():
        y = op.forward(x, res=res, in_act=ia, out_act=out_act, mask=mask, out_mask=out_mask)       # ([B, C_out, T_out]: an even k shortens the sequence)
        return y, op.kernel_instance()

    vs_option("VS_NO_KTAP", 1)
    y_ref, k_ref = run()
    vs_option("VS_NO_KTAP", 0)
    y_new, k_new = run()
    tile = {"conv_split_kernel<1, 4, 2, 2, %d>" % (3 if s3 else 1): "2, 2, 4", "conv_split_kernel<1, 2, 1, 4, %d>" % (3 if s3 else 1): "1, 4, 2",
            "conv_split_kernel<1, 1, 1, 4, %d>" % (3 if s3 else 1): "1, 4, 1"}
    assert k_ref in tile, k_ref
    assert k_new == "conv_ktap_kernel<%d, %d, %d, 0, %s, 1>" % (k, in_act, 2 if s3 else 1, tile[k_ref]), (k_ref, k_new)
    # (round 5 exempted plain bf16 behind a masked input: the tile kernel's instances gave run-to-run different results there.  Round 6 root-caused it -- a
    #  packed multiply with op_sel on its second source, wrong in lanes 48-63 beside MFMAs: conv_common.h, tests/test_conv_mask_race_gpu.py -- and the
    #  exemption is gone)
    assert torch.equal(y_new, y_ref), float((y_new - y_ref).abs().max())
    y2, _ = run()
    assert torch.equal(y2, y_new)
    n = min(1024, y_new.shape[2])
    xs = x[:1].double()
    if mask is not None and in_act >= 2:
        xs = xs * mask[:1, None].double()
    wf = w.double().transpose(0, 1).flip(2) if adj else w.double()
    ref = torch.nn.functional.conv1d(xs, wf, None if bias is None else bias.double(), padding=(k * d - d) // 2, dilation=d)[:, :, :n]
    if use_res:
        ref = ref + res[:1, :, :n].double()
    ref = torch.tanh(ref) if out_act == 1 else (torch.relu(ref) if out_act == 2 else ref)
    if out_mask:
        ref = ref * mask[:1, None, :n].double()
    err = (y_new[:1, :, :n].double() - ref)
    assert float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) <= (2e-6 if s3 else 1e-2), float(err.abs().max())


@pytest.mark.parametrize("math,H,gin,B,T,mode", [(3, 192, 256, 32, 1024, "gate"), (3, 192, 0, 3, 700, "gate"), (1, 512, 256, 8, 4096, "gate"), (3, 96, 0, 4, 1024, "coupling_inv")])
def test_ktap_paired_instances_are_bit_identical_to_the_tile_kernel(math, H, gin, B, T, mode, vs_option):
    """VS_CONV1D_PAIRED (csrc/conv_ktap_pair.hip): the WaveNet's k = 5 in_layers with the gate in the epilogue (reference modules/visinger/encoder.py:158-161, 206-213:
    tanh(a + g_a) * sigmoid(b + g_b) over the row pair (c, H + c), the conditioning as a per-item bias) -- and the coupling update of the flow's `post` conv shape
    (flow.py:66-85) -- against conv_split_kernel<2, 2, 2, 2, T>, bit for bit, and against fp64"""
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    vs_option("VS_CONV_MATH", math)
    g = torch.Generator(device="cuda").manual_seed(4000 + H + T)
    op = ConvOp(L.CONV1D_PAIRED, H, 2 * H, 5, 1, 2)
    w = torch.randn(2 * H, H, 5, device="cuda", generator=g) * (H * 5) ** -0.5
    bias = torch.randn(2 * H, device="cuda", generator=g) * 0.1
    op.set_weights(w, None, bias)
    x = torch.randn(B, H, T, device="cuda", generator=g)
    x[:, : H // 2] *= torch.exp2(torch.randint(-5, 6, (B, H // 2, 1), device="cuda", generator=g).float())
    bias_b = torch.randn(B, 2 * H, device="cuda", generator=g) if gin else None
    lens = torch.randint(T // 2, T + 1, (B,), device="cuda", generator=g)
    lens[0] = T
    mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).float()
    x1 = torch.randn(B, H, T, device="cuda", generator=g) if mode != "gate" else None

    def run():
        if mode == "gate":
            y = op.forward(x, bias_b=bias_b, pair_mode=L.PAIR_GATE)
        else:
            y = op.forward(x, res=x1, mask=mask, pair_mode=L.PAIR_COUPLING_INV)
        return y, op.kernel_instance()

    vs_option("VS_NO_KTAP", 1)
    y_ref, k_ref = run()
    vs_option("VS_NO_KTAP", 0)
    y_new, k_new = run()
    assert k_ref == "conv_split_kernel<2, 2, 2, 2, %d>" % math, k_ref
    assert k_new == "conv_ktap_kernel<5, 0, %d, 0, 2, 2, 2, 2>" % (2 if math == 3 else 1), k_new
    assert torch.equal(y_new, y_ref), float((y_new - y_ref).abs().max())
    y2, _ = run()
    assert torch.equal(y2, y_new)
    z = torch.nn.functional.conv1d(x[:1].double(), w.double(), bias.double(), padding=2)
    if bias_b is not None:
        z = z + bias_b[:1, :, None].double()
    a, b_ = z[:, :H], z[:, H:]
    if mode == "gate":
        ref = torch.tanh(a) * torch.sigmoid(b_)
    else:
        m = mask[:1, None].double()
        ref = (x1[:1].double() - a * m) * torch.exp(-b_ * m) * m
    err = y_new[:1].double() - ref
    assert float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) <= (3e-6 if math == 3 else 1e-2), float(err.abs().max())


def test_ktap_kernel_dispatch_and_fallbacks(vs_option):
    """taken for plain stride-1 convs of 3 / 7 / 11 taps on whole 16-channel chunks; everything else stays on the tile kernel"""
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    vs_option("VS_CONV_MATH", 3)
    vs_option("VS_NO_SMALL_GRID", 1)

    def inst(cin, cout, k, T=8192, B=4):
        op = ConvOp(L.CONV1D, cin, cout, k, 1, k // 2)
        op.set_weights(torch.randn(cout, cin, k, device="cuda") * 0.03, None, torch.zeros(cout, device="cuda"))
        op.forward(torch.randn(B, cin, T, device="cuda"))
        return op.kernel_instance()

    assert inst(128, 128, 7) == "conv_ktap_kernel<7, 0, 2, 0, 4, 1, 8, 1>"
    assert inst(128, 128, 9) == "conv_ktap_kernel<9, 0, 2, 0, 4, 1, 8, 1>"
    assert inst(128, 128, 5) == "conv_split_kernel<1, 8, 4, 1, 3>"              # no instance for 5 taps
    assert inst(120, 128, 7) == "conv_split_kernel<1, 8, 4, 1, 3>"              # C_in not a multiple of 16
    vs_option("VS_NO_KTAP", 1)
    assert inst(128, 128, 7) == "conv_split_kernel<1, 8, 4, 1, 3>"


TR_CASES = [
    # C_in, C_out, k, stride, pad, B, T_in, in_act, acc  -- the generator's upsamplers (decoder.py:36-48: ConvTranspose1d(k = 2 * stride, padding = (k - stride) / 2) behind a leaky-relu)
    (512, 256, 16, 8, 4, 4, 1024, 1, False),        # stage 1 of the hop-256 generator
    (256, 128, 16, 8, 4, 2, 2048 + 37, 1, False),   # stage 2, ragged last tile (the element-wise polyphase stores)
    (128, 64, 4, 2, 1, 3, 8192, 1, False),          # stage 3: stride 2, 128 virtual rows = one row block
    (256, 128, 8, 4, 2, 2, 3000, 1, False),         # stride 4, T_in not a multiple of the tile
]


@pytest.mark.parametrize("case", TR_CASES, ids=lambda c: "c%d-%d_k%d_s%d_B%d_T%d_a%d" % (c[0], c[1], c[2], c[3], c[5], c[6], c[7]))
def test_transposed_ktap_instance_is_bit_identical_to_the_tile_kernel(case, vs_option):
    """conv_ktap_kernel<2, ACT, 2, 4, 4, 1, 8, 1> (conv_ktap.inc IO bit 2: a transposed conv whose every phase uses two of the three packed taps; per-wave first
    tap, polyphase epilogue) against conv_split_tr_kernel<1, 8, 4, 1, 3>: same planes, same MFMA order (the skipped products are exact zeros), same stores."""
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    cin, cout, k, st, pad, B, T, in_act, use_acc = case
    vs_option("VS_CONV_MATH", 3)
    vs_option("VS_NO_SMALL_GRID", 1)
    g = torch.Generator(device="cuda").manual_seed(77 + k + st + cin)
    op = ConvOp(L.CONV_TRANSPOSE1D, cin, cout, k, st, pad)
    w = torch.randn(cin, cout, k, device="cuda", generator=g) * (cin * k / st) ** -0.5
    bias = torch.randn(cout, device="cuda", generator=g) * 0.1
    op.set_weights(w, None, bias)
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    x[:, : cin // 2] *= torch.exp2(torch.randint(-6, 7, (B, cin // 2, 1), device="cuda", generator=g).float())
    x[0, 5, 100:400] = 2.0 ** 9
    Tout = op.out_len(T)
    acc = torch.randn(B, cout, Tout, device="cuda", generator=g) if use_acc else None
    ia = (L.IN_NONE, L.IN_LRELU)[in_act]

    def run():
        y = torch.empty((B, cout, Tout), device="cuda")
        op.forward(x, y=y, acc=acc, in_act=ia)
        return y, op.kernel_instance()

    vs_option("VS_NO_KTAP", 1)
    y_ref, k_ref = run()
    vs_option("VS_NO_KTAP", 0)
    y_new, k_new = run()
    assert k_ref == "conv_split_tr_kernel<1, 8, 4, 1, 3>" and k_new == "conv_ktap_kernel<2, %d, 2, 4, 4, 1, 8, 1>" % in_act, (k_ref, k_new)
    assert torch.equal(y_new, y_ref), float((y_new - y_ref).abs().max())
    assert torch.equal(run()[0], y_new)
    xs = x[:1].double()
    xs = torch.where(xs > 0, xs, 0.1 * xs) if in_act == 1 else xs
    ref = torch.nn.functional.conv_transpose1d(xs, w.double(), bias.double(), stride=st, padding=pad)
    if use_acc:
        ref = ref + acc[:1].double()
    err = float((y_new[:1].double() - ref).abs().max())
    assert err <= 3e-6 * float(ref.abs().max()) + 1e-6, err
