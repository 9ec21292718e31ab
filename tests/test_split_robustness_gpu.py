"""Robustness of the split-bf16 x6 arithmetic (csrc/conv_split.hip, conv_common.h split_pair) beyond N(0, 1) data: operands spanning
forty binary orders of magnitude, catastrophic cancellation, and non-finite inputs -- against the fp64 oracle and next to the
exact-fp32 MFMA engine on the same data.  Reference arithmetic: torch's fp32 conv (config/models/base_config.yaml:5 `amp: false`).

Error measure: |y - y_fp64| against S = sum |x| |w| over the receptive field (the condition-aware scale of a dot product: an fp32
evaluation in any order is bounded by ~n * 2^-24 * S and typically sits at sqrt(n) * 2^-24 * S; a bound relative to |y| is
meaningless under cancellation)."""
import numpy as np
import pytest
import torch

from visinger_amd import _lib as L

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def run(oracle, x, w, bias, k, d, maths=(L.MATH_F32, L.MATH_SPLIT6, L.MATH_SPLIT3)):
    from visinger_amd.ops import ConvOp
    C_out, C_in, _ = w.shape
    pad = d * (k - 1) // 2
    ref = oracle.conv1d(x.astype(np.float64), w, bias, dilation=d, padding=pad)
    S = oracle.conv1d(np.abs(x).astype(np.float64), np.abs(w), None if bias is None else np.abs(bias), dilation=d, padding=pad)
    out = {}
    for math in maths:
        op = ConvOp(L.CONV1D, C_in, C_out, k, d, pad).set_math(math)
        op.set_weights(dev(w), None, None if bias is None else dev(bias))
        y = op.forward(dev(x)).cpu().double().numpy()
        out[math] = (y, float(np.max(np.abs(y - ref) / np.maximum(S, 1e-300))), float(np.sqrt(np.mean(((y - ref) / np.maximum(S, 1e-300)) ** 2))))
    return ref, S, out


@pytest.mark.parametrize("C,k,d,T,span", [(128, 7, 1, 1500, 20), (64, 11, 3, 2048, 20), (256, 3, 1, 700, 12), (32, 7, 5, 3000, 20)])
def test_wide_dynamic_range(oracle, vs_option, C, k, d, T, span):
    """activations and weights with magnitudes log-uniform over 2^-span .. 2^span and random signs: every plane of the split sees
    every exponent; the lower planes (x - xh, x - xh - xm) of large and of tiny values are all exercised"""
    vs_option("VS_NO_WINO", 1)
    r = np.random.default_rng(C + 13 * k + d)
    x = (r.choice([-1.0, 1.0], (2, C, T)) * np.exp2(r.uniform(-span, span, (2, C, T))) * r.uniform(1, 2, (2, C, T))).astype(np.float32)
    w = (r.choice([-1.0, 1.0], (C, C, k)) * np.exp2(r.uniform(-span, span, (C, C, k))) * r.uniform(1, 2, (C, C, k))).astype(np.float32)
    bias = r.standard_normal(C).astype(np.float32)
    _, _, out = run(oracle, x, w, bias, k, d)
    (_, max32, rms32), (_, max6, rms6) = out[L.MATH_F32], out[L.MATH_SPLIT6]
    n = C * k
    print(f"wide range C={C} k={k} d={d}: max|err|/S split6 {max6:.3e} fp32-mfma {max32:.3e}; rms split6 {rms6:.3e} fp32-mfma {rms32:.3e}")
    assert max6 <= 4 * 2.0 ** -24 * np.sqrt(n) + 2.0 ** -22, (max6, max32)      # fp32 class: a few ulp-of-S, like the fp32 engine
    # sums dominated by a handful of huge products: the split's dropped cross terms (<= 3 * 2^-24 of a product, one-sided because the
    # planes are truncations) weigh as much as the fp32 engine's own product rounding (<= 2^-24): same class, up to ~3x its rms
    assert rms6 <= 3.0 * rms32 + 2.0 ** -26, (rms6, rms32)
    # split-f16 x3: one power-of-two scale per staged tile (16 channels x ~320 positions) and per conv weight -- operands more than 2^17
    # below their tile's largest magnitude lose relative precision (absolute floor 2^-39 of that magnitude), which a sum normalised by
    # S = sum |x||w| does not see: the same bounds hold
    _, max3, rms3 = out[L.MATH_SPLIT3]
    print(f"                              split3 max {max3:.3e} rms {rms3:.3e}")
    assert max3 <= 4 * 2.0 ** -24 * np.sqrt(n) + 2.0 ** -22, (max3, max32)
    assert rms3 <= 3.0 * rms32 + 2.0 ** -26, (rms3, rms32)


@pytest.mark.parametrize("C,k,T", [(128, 7, 1024), (64, 3, 4096)])
def test_catastrophic_cancellation(oracle, vs_option, C, k, T):
    """pairs of channels carry +a and -a(1 - 2^-12) under equal weights: each output is the small difference of products 2^12
    times larger, so any error of the operand split relative to the OPERANDS (not to the result) would surface 4096-fold"""
    vs_option("VS_NO_WINO", 1)
    r = np.random.default_rng(7 * C + k)
    a = (r.standard_normal((2, C // 2, T)) * 100.0).astype(np.float32)
    x = np.empty((2, C, T), np.float32)
    x[:, 0::2] = a
    x[:, 1::2] = -(a * np.float32(1.0 - 2.0 ** -12)).astype(np.float32)
    w = np.repeat((r.standard_normal((C, C // 2, k)) / np.sqrt(C * k)).astype(np.float32), 2, axis=1)      # equal weights within a pair
    ref, S, out = run(oracle, x, w, None, k, 1)
    assert np.median(np.abs(ref) / S) < 2e-4                       # the data really cancels
    (_, max32, rms32), (_, max6, rms6) = out[L.MATH_F32], out[L.MATH_SPLIT6]
    print(f"cancellation C={C} k={k}: max|err|/S split6 {max6:.3e} fp32-mfma {max32:.3e}; rms split6 {rms6:.3e} fp32-mfma {rms32:.3e}")
    assert max6 <= 4 * 2.0 ** -24 * np.sqrt(C * k) + 2.0 ** -22, (max6, max32)
    assert rms6 <= 1.5 * rms32 + 2.0 ** -26, (rms6, rms32)
    # split-f16 x3 represents an operand with 22 significant bits (round to nearest on both planes: <= 2^-23 of the OPERAND, rms ~2^-24.3
    # -- the size of one fp32 rounding of it); this test surfaces exactly that, 4096-fold relative to the result: stated bound 3x the
    # fp32 engine's rms error relative to S, same max bound
    _, max3, rms3 = out[L.MATH_SPLIT3]
    print(f"                              split3 max {max3:.3e} rms {rms3:.3e}")
    assert max3 <= 4 * 2.0 ** -24 * np.sqrt(C * k) + 2.0 ** -22, (max3, max32)
    assert rms3 <= 3.0 * rms32 + 2.0 ** -26, (rms3, rms32)
    # alternating signs with exact cancellation: sum_k (+v, -v) * 1 == 0 exactly in every arithmetic (the six cross products of
    # +v and -v cancel pairwise inside the accumulator: no residue from the split)
    x2 = np.empty((1, C, T), np.float32)
    v = (r.standard_normal((1, C // 2, T)) * np.exp2(r.integers(-20, 20, (1, C // 2, T)))).astype(np.float32)
    x2[:, 0::2], x2[:, 1::2] = v, -v
    w2 = np.ones((C, C, 1), np.float32)
    _, _, out2 = run(oracle, x2, w2, None, 1, 1)
    for math, (y, _, _) in out2.items():
        assert np.abs(y).max() <= 2.0 ** -24 * np.abs(v).max() * C, math


def test_nonfinite_inputs_stay_local_and_nonfinite(oracle):
    """A +-Inf or NaN activation: the reference (torch fp32 conv) returns +-Inf or NaN in the outputs whose receptive field holds
    it.  split_pair's lower planes of an Inf are Inf - Inf = NaN, so the split engine returns NaN where the reference may return
    +-Inf: outputs are non-finite in exactly the same positions, every other output is untouched.  ('non-finite')"""
    from visinger_amd.ops import ConvOp
    r = np.random.default_rng(3)
    C, k, d, T = 64, 7, 3, 1024
    pad = d * (k - 1) // 2
    x = r.standard_normal((2, C, T)).astype(np.float32)
    w = (r.standard_normal((C, C, k)) / np.sqrt(C * k)).astype(np.float32)
    clean = oracle.conv1d(x.astype(np.float64), w, None, dilation=d, padding=pad)
    x[0, 5, 100], x[0, 17, 600], x[1, 63, 1023] = np.inf, -np.inf, np.nan
    hit = np.zeros((2, T), bool)
    for b, t in ((0, 100), (0, 600), (1, 1023)):
        for j in range(k):
            tt = t + pad - j * d
            if 0 <= tt < T:
                hit[b, tt] = True
    for math in (L.MATH_SPLIT6, L.MATH_SPLIT3, L.MATH_F32):
        op = ConvOp(L.CONV1D, C, C, k, d, pad).set_math(math)
        op.set_weights(dev(w), None, None)
        y = op.forward(dev(x)).cpu().double().numpy()
        bad = ~np.isfinite(y)
        assert (bad == hit[:, None, :]).all(), math                 # non-finite exactly in the receptive fields, for every row
        assert np.abs(y[~bad] - np.broadcast_to(clean, y.shape)[~bad]).max() <= 2e-5, math


@pytest.mark.parametrize("C,k,T", [(128, 7, 2048), (64, 3, 4096)])
def test_outlier_channel_inside_a_chunk_costs_the_small_values_their_low_bits(oracle, vs_option, C, k, T):
    """ONE channel of a 16-channel chunk carries values ~1e7 (2^23) times its neighbours', and the outputs under test
    take (almost) nothing from it -- the outlier's weights are zero for the first half of the output rows, so those rows are sums over the
    SMALL channels only.  The split-f16 arithmetic scales a staged 16-channel tile by ONE power of two that puts the outlier below 2^15:
    the small values then sit near 2^-8 of the scaled range, where their two f16 planes keep an ABSOLUTE precision of 2^-25 of the scaled
    unit (f16 subnormal spacing of the low plane) = about 2^-16 .. 2^-17 RELATIVE to the small values themselves.  The S-normalised bounds
    of the tests above cannot see this (S is dominated by the outlier); here the error is measured relative to the SMALL values' own
    S_small = sum |x_small| |w|.  Stated behaviour: the split-f16 engine is fp32-class (<= 2^-21) wherever a tile's dynamic range stays
    below 2^17 and degrades gracefully to <= 2^-14 of S_small at a 2^23 range inside one chunk (measured ~2^-16); the exact-fp32 MFMA engine
    and the split-bf16 x6 engine (8-bit exponents per plane, no tile scale) keep <= 2^-21 on the same data.  The rows that DO take the
    outlier stay fp32-class relative to their S."""
    vs_option("VS_NO_WINO", 1)
    r = np.random.default_rng(31 * C + k)
    x = r.standard_normal((2, C, T)).astype(np.float32)
    out_ch = 5                                                   # inside the first 16-channel chunk
    x[:, out_ch] *= np.float32(1.0e7)
    w = (r.standard_normal((C, C, k)) * (C * k) ** -0.5).astype(np.float32)
    w[: C // 2, out_ch, :] = 0.0                                 # the first half of the rows never sees the outlier channel
    ref, S, out = run(oracle, x, w, None, k, 1)
    half = C // 2
    for math, name, bar in ((L.MATH_F32, "fp32-mfma", 2.0 ** -21), (L.MATH_SPLIT6, "split6", 2.0 ** -21), (L.MATH_SPLIT3, "split3", 2.0 ** -14)):
        y = out[math][0]
        e_small = float(np.max(np.abs(y[:, :half] - ref[:, :half]) / S[:, :half]))          # S of these rows = S_small (outlier weights are 0)
        e_big = float(np.max(np.abs(y[:, half:] - ref[:, half:]) / S[:, half:]))
        print(f"outlier channel C={C} k={k}: {name:9s} rows without the outlier max|err|/S_small {e_small:.3e} (2^{np.log2(max(e_small, 1e-30)):.1f}); "
              f"rows with it max|err|/S {e_big:.3e}")
        assert e_small <= bar, (name, e_small)
        assert e_big <= 4 * 2.0 ** -24 * np.sqrt(C * k) + 2.0 ** -22, (name, e_big)        # (S-normalised, as in test_wide_dynamic_range: every engine is fp32-class there)
    # and the loss is what the arithmetic's description says, not more: the small values keep at least 14 bits next to a 2^23 outlier
    assert float(np.max(np.abs(out[L.MATH_SPLIT3][0][:, :half] - ref[:, :half]) / S[:, :half])) > 2.0 ** -24      # (it IS visible: this test can see it)


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Trained-like statistics at MODEL level: every model-level check so far ran random-init weights with O(1) activations.
# A HiFi-GAN generator is equivariant under positive per-channel gains (leaky-relu is positively homogeneous): scale the rows that PRODUCE a channel by g and
# the columns that CONSUME it by 1 / g and the function is unchanged -- but every intermediate tensor and every weight now carries the gains.  The residual
# stream of a stage shares one gain vector (x = xt + x ties its channels across the stage's nine pairs); the channel between the two convs of a pair has its
# own.  decoder.py:40-59, 91-104.


def _effective(conv):
    from visinger_amd.modules.hipconv import _HipConvMixin                    # noqa: F401
    w, g = conv._weights()
    w = w.detach().double()
    if g is not None:
        w = w * (g.detach().double().reshape(-1, 1, 1) / w.flatten(1).norm(dim=1).reshape(-1, 1, 1))
    return w


def _set_effective(conv, w, bias_gain=None):
    """make `w` (fp64) the conv's effective weight: weight_v = w, weight_g = its row norms (weight norm folds back to w up to one fp32 rounding)"""
    with torch.no_grad():
        if hasattr(conv, "weight_g"):
            conv.weight_v.copy_(w.float())
            conv.weight_g.copy_(conv.weight_v.flatten(1).norm(dim=1).reshape(conv.weight_g.shape))
        else:
            conv.weight.copy_(w.float())
        if bias_gain is not None and conv.bias is not None:
            conv.bias.mul_(bias_gain.float())


def apply_channel_gains(gen, draw):
    """draw(n) -> n positive gains.  Rewrites the generator's parameters in place; the function it computes is unchanged in exact arithmetic."""
    nk = gen.num_kernels
    g_in = draw(gen.conv_pre.out_channels).double()                                         # the 512-channel tensor between conv_pre (+ cond) and ups[0]
    _set_effective(gen.conv_pre, _effective(gen.conv_pre) * g_in.reshape(-1, 1, 1), g_in)
    if hasattr(gen, "cond"):
        _set_effective(gen.cond, _effective(gen.cond) * g_in.reshape(-1, 1, 1), g_in)
    for i, up in enumerate(gen.ups):
        s = draw(up.out_channels).double()                                                  # this stage's residual stream
        _set_effective(up, _effective(up) * s.reshape(1, -1, 1) / g_in.reshape(-1, 1, 1), s)   # ConvTranspose1d weight: [C_in, C_out, k]
        for block in gen.resblocks[i * nk:(i + 1) * nk]:
            for c1, c2 in zip(block.convs1, block.convs2):
                h = draw(c1.out_channels).double()                                          # the channel between the two convs of a pair
                _set_effective(c1, _effective(c1) * h.reshape(-1, 1, 1) / s.reshape(1, -1, 1), h)
                _set_effective(c2, _effective(c2) * s.reshape(-1, 1, 1) / h.reshape(1, -1, 1), s)
        g_in = s
    _set_effective(gen.conv_post, _effective(gen.conv_post) / g_in.reshape(1, -1, 1))


def _generator_case(oracle, draw, select, capsys, label):
    import bench
    from conftest import usable_cores
    from visinger_amd.modules.hipconv import select_math_by_weight_range
    model, hp = bench.build_model()
    gen = model.decoder
    with torch.no_grad():        # trained-like weight norms on top: per-row g log-uniform over 2^-2 .. 2^2 (before the gains)
        gq = torch.Generator().manual_seed(11)
        for m in gen.modules():
            if hasattr(m, "weight_g"):
                m.weight_g.mul_(torch.exp2(torch.rand(m.weight_g.shape, generator=gq) * 4 - 2))
    B, T = 8, 512                                                                           # BASELINE configs[1]'s size: the production dispatch
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, 192, T, generator=g)
    spk = torch.randn(B, 256, 1, generator=g) * 0.1
    sd0 = {k: v.detach().cpu().numpy().copy() for k, v in gen.state_dict().items()}
    apply_channel_gains(gen, draw)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in gen.state_dict().items()}
    kw = dict(resblock=hp["dec_blocks"], resblock_kernel_sizes=hp["dec_kernel_size"], resblock_dilation_sizes=hp["dec_dilation_sizes"],
              upsample_rates=hp["upsample_rates"], upsample_kernel_sizes=hp["upsample_kernel_sizes"])
    oracle.set_threads(usable_cores())
    items = [0, B - 1]
    ref = oracle.generator(sd, z[items].numpy(), spk[items].numpy(), **kw)[:, 0]
    ref0 = oracle.generator(sd0, z[items[:1]].numpy(), spk[items[:1]].numpy(), **kw)[:, 0]
    same_function = float(np.abs(ref[:1] - ref0).max())                                    # the gains leave the function alone (fp64 referee on both; item 0)
    gen = gen.cuda().eval()
    switched = select_math_by_weight_range(gen) if select else []
    with torch.no_grad():
        wav = gen(z.cuda(), g=spk.cuda()).squeeze(1)
    torch.cuda.synchronize()
    got = wav[items].cpu().double().numpy()
    e = np.abs(got - ref)
    with capsys.disabled():
        print(f"\n   generator, B={B} T_mel={T}, {label}: waveform max / rms err vs fp64 {e.max():.2e} / {np.sqrt((e ** 2).mean()):.2e} "
              f"(rms of the waveform {np.sqrt((ref ** 2).mean()):.3f}); gained vs plain parameters in fp64: {same_function:.1e}; "
              f"convs switched to the exact bf16 x3 split: {len(switched)}" + (f" (largest row drop {max(d for _, d in switched):.1f} bits)" if switched else ""))
    assert same_function <= 1e-6
    assert np.isfinite(got).all()
    return float(e.max()), switched


def test_generator_with_trained_like_channel_gains(oracle, capsys):
    """per-channel gains log-uniform over 2^-4 .. 2^4 on every tensor of the generator (activations of one 16-channel tile differ by up to 2^8, weights of one
    conv by up to 2^16 on top of their own spread): the split-f16 arithmetic as dispatched in production, no fallback -- waveform <= 1e-4 abs (north_star)"""
    gq = torch.Generator().manual_seed(21)
    emax, switched = _generator_case(oracle, lambda n: torch.exp2(torch.rand(n, generator=gq) * 8 - 4), False, capsys, "gains 2^-4 .. 2^4, split-f16 everywhere")
    assert emax <= 1e-4


def test_generator_with_an_outlier_channel_in_every_tile(oracle, capsys):
    """one channel of every 16-channel tile 2^12 above its neighbours, in every tensor of the generator (the per-tile scale is set by the outlier: its
    neighbours keep 2^-12 of the tile's range): split-f16 as dispatched -- waveform <= 1e-4 abs"""
    def draw(n):
        gains = torch.ones(n)
        gains[3::16] = 4096.0
        return gains
    emax, switched = _generator_case(oracle, draw, False, capsys, "one channel in 16 at 2^12, split-f16 everywhere")
    assert emax <= 1e-4


def test_generator_with_extreme_gains_through_the_weight_range_check(oracle, capsys):
    """gains log-uniform over 2^-10 .. 2^10: rows of one conv now differ by up to 2^20 and elements by up to 2^40 -- beyond what ONE scale per conv can carry
    in two f16 planes (a weight 2^-40 below the largest flushes to zero while the activation it multiplies is 2^20 above its tile's others).
    hipconv.select_math_by_weight_range (the load-time check INTEGRATION.md 4 prescribes for real checkpoints) moves exactly those convs to the exact
    bf16 x3 split; with it the waveform is <= 1e-4 abs.  Without it the error is reported (not asserted: it is the documented limit of the arithmetic)."""
    gq = torch.Generator().manual_seed(22)
    draw = lambda n: torch.exp2(torch.rand(n, generator=gq) * 20 - 10)
    emax, switched = _generator_case(oracle, draw, True, capsys, "gains 2^-10 .. 2^10, weight-range check on")
    assert switched and emax <= 1e-4
    gq.manual_seed(22)
    emax_raw, _ = _generator_case(oracle, draw, False, capsys, "gains 2^-10 .. 2^10, split-f16 everywhere (the documented limit)")
    assert np.isfinite(emax_raw)
