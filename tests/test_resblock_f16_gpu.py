"""csrc/resblock_f16.hip: a whole MRF residual block (reference modules/visinger/decoder.py:91-104) as ONE launch on the split-f16 x3
arithmetic, against the fp64 oracle's ResBlock1: every kernel size of the generator and two more, both widths, the whole block and pair by
pair, tiles in the interior / at both ends of the sequence / a sequence shorter than one tile / lengths that are no multiple of 4
(element-wise epilogue), the MRF accumulate input and scale, non-finite inputs, and the dispatch of ResBlock1 itself."""
import os

import numpy as np
import pytest
import torch

from visinger_amd import _lib as L

pytestmark = pytest.mark.gpu


def _block(C, k, seed):
    from visinger_amd.modules.hipconv import set_conv_math
    from visinger_amd.modules.visinger.decoder import ResBlock1
    torch.manual_seed(seed)
    m = ResBlock1(C, k, (1, 3, 5))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("weight_g"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / (p.shape[1] * p.shape[2]) ** 0.5)
    sd = {k_: v.detach().numpy().copy() for k_, v in m.state_dict().items()}
    return set_conv_math(m.cuda().eval(), L.MATH_SPLIT3), sd


@pytest.mark.parametrize("C,k,B,T", [(32, 3, 2, 3000), (64, 3, 2, 1501), (32, 7, 1, 2048), (64, 7, 2, 777), (32, 11, 2, 1000), (64, 11, 1, 4096),
                                     (64, 5, 1, 100), (32, 3, 3, 7), (64, 9, 1, 232), (32, 5, 1, 233), (128, 3, 2, 1000), (128, 7, 1, 515),
                                     (128, 11, 1, 300), (32, 7, 2, 5000),
                                     # (round 6: 64 channels with a wide halo on launches that fill the chip twice over -> 512-column tiles on eight waves)
                                     (64, 11, 16, 14000), (64, 7, 24, 10001)])
@pytest.mark.parametrize("pairs", [3, 1])
def test_whole_resblock_launch_vs_oracle(oracle, vs_option, C, k, B, T, pairs):
    m, sd = _block(C, k, C + k)
    vs_option("VS_RESBLOCK_PAIRS", pairs)
    r = np.random.default_rng(C * 7 + k + T)
    x = (2.0 * r.standard_normal((B, C, T))).astype(np.float32)
    acc = r.standard_normal((B, C, T)).astype(np.float32)
    ref = oracle.resblock1(sd, x.astype(np.float64), kernel_size=k, dilation=(1, 3, 5))
    scale = float(np.sqrt((ref ** 2).mean()))
    xd = torch.from_numpy(x).cuda()
    with torch.no_grad():
        y = m._run_fused(xd, torch.empty_like(xd), first=True, scale=1.0)
        acc_t = torch.from_numpy(acc).cuda()
        m._run_fused(xd, acc_t, first=False, scale=1.0 / 3.0)                  # in place on the MRF accumulator, averaged
    assert m.convs1[0]._op().kernel_instance().startswith("resblock_f16_kernel<"), m.convs1[0]._op().kernel_instance()
    if C == 64 and B >= 16:
        assert m.convs1[2]._op().kernel_instance().startswith("resblock_f16_kernel<4, 2, 4,"), m.convs1[2]._op().kernel_instance()      # the last pair / the whole block: H >= 20
    assert np.abs(y.double().cpu().numpy() - ref).max() <= 1e-5 * scale
    assert np.abs(acc_t.double().cpu().numpy() - (ref + acc) / 3.0).max() <= 1e-5 * scale
    assert torch.equal(xd.cpu(), torch.from_numpy(x))                            # the input is left untouched


def test_whole_resblock_equals_conv_by_conv_launches(vs_option):
    """the fused launch against the SAME arithmetic launched conv by conv (csrc/conv_split.hip, TERMS = 3) at a production-sized tile count:
    identical products and scales per tile differ (a staged chunk tile vs the whole-channel tile), so agreement is to fp32 rounding"""
    m, _ = _block(64, 7, 5)
    x = torch.randn(4, 64, 16384, device="cuda")
    with torch.no_grad():
        y_f = m._run_fused(x, torch.empty_like(x)).clone()
        vs_option("VS_NO_RESBLOCK_FUSED", 1)
        y_u = m._run_fused(x, torch.empty_like(x))
    assert m.convs1[0]._op().kernel_instance().startswith(("conv_split_kernel<", "conv_ktap_kernel<"))       # (round 5: the 64 x 256 conv_ktap instance, bit-identical to the tile kernel)
    assert float((y_f - y_u).abs().max()) <= 2e-6 * float(y_u.abs().max())


def test_resblock_nonfinite_inputs_stay_local():
    """an Inf / NaN activation poisons exactly the outputs whose receptive field (through all six convs) holds it: the tile's scale is taken
    from its FINITE magnitudes (conv_common.h f16_maxkey), so every other output keeps its value"""
    m, _ = _block(32, 3, 9)
    x = torch.randn(1, 32, 4096, device="cuda")
    with torch.no_grad():
        clean = m._run_fused(x, torch.empty_like(x)).clone()
        x[0, 5, 1000], x[0, 20, 3000] = float("inf"), float("nan")
        y = m._run_fused(x, torch.empty_like(x))
    bad = ~torch.isfinite(y[0]).all(0)
    H = 12                                       # pads of the six convs: (1 + 1) + (3 + 1) + (5 + 1)
    want = torch.zeros(4096, dtype=torch.bool, device="cuda")
    for t in (1000, 3000):
        want[t - H:t + H + 1] = True
    assert bool((bad <= want).all()) and bool(bad[1000]) and bool(bad[3000])       # nothing outside the receptive cone
    ok = ~want
    assert float((y[0][:, ok] - clean[0][:, ok]).abs().max()) <= 1e-5 * float(clean.abs().max())


def _bf16_emulation(m, x, acc=None, scale=1.0, pairs=3):
    """resblock_bf16_kernel's arithmetic restated with torch fp64 convs: leaky-relu in fp32, operands rounded to bf16 (RNE), exact products,
    residual stream and bias in fp32 registers (here fp64: the difference is the fp32 accumulation order), ONE rounding to bf16 at the end
    of a launch (`pairs` pairs per launch: the tensors between launches are bf16-resident)"""
    import torch.nn.functional as F
    bf = lambda t: t.float().bfloat16().double()
    cur = x.double()
    n = len(m.convs1)
    for i, (c1, c2) in enumerate(zip(m.convs1, m.convs2)):
        def w(c):
            from visinger_amd.ops import weightnorm_fold
            return bf(weightnorm_fold(c.weight_v.detach(), c.weight_g.detach()))       # the fp32 weight the library rounds (a12)
        xt = F.conv1d(bf(F.leaky_relu(cur.float(), 0.1)), w(c1), c1.bias.double(), padding=c1.padding[0], dilation=c1.dilation[0])
        xt = F.conv1d(bf(F.leaky_relu(xt.float(), 0.1)), w(c2), c2.bias.double(), padding=c2.padding[0])
        cur = xt + cur
        if (i + 1) % pairs == 0 and i + 1 < n:
            cur = bf(cur)
    if acc is not None:
        cur = cur + acc.double()
    return cur * scale


@pytest.mark.parametrize("C,k,B,T", [(32, 3, 2, 3000), (64, 3, 2, 1501), (32, 7, 1, 2048), (64, 7, 2, 777), (32, 11, 2, 1000), (64, 11, 1, 1024),
                                     (32, 3, 3, 7), (64, 9, 1, 232), (128, 3, 2, 1000), (128, 7, 1, 515), (128, 11, 1, 300), (64, 7, 16, 14000)])
@pytest.mark.parametrize("pairs", [3, 1])
def test_bf16_resident_resblock_launch_vs_its_arithmetic(vs_option, C, k, B, T, pairs):
    """resblock_bf16_kernel (VS_MATH_BF16 on bf16-RESIDENT tensors, BASELINE configs[4]): the launch against an fp64 restatement of exactly
    its arithmetic (bf16 roundings of operands, weights as the library folds them, bf16 tensors between launches, one final rounding).
    What may differ is the order of the fp32 accumulation, i.e. values that sit on a bf16 rounding boundary."""
    from visinger_amd.modules.hipconv import set_conv_math
    m, _ = _block(C, k, C + k)
    set_conv_math(m, L.MATH_BF16)
    vs_option("VS_RESBLOCK_PAIRS", pairs)
    g = torch.Generator(device="cuda").manual_seed(C * 7 + k + T)
    x = (2.0 * torch.randn(B, C, T, device="cuda", generator=g)).bfloat16()
    acc = torch.randn(B, C, T, device="cuda", generator=g).bfloat16()
    x0 = x.clone()
    with torch.no_grad():
        y = m._run_fused(x, torch.empty_like(x), first=True, scale=1.0)
        name = m.convs1[0]._op().kernel_instance()
        acc_t = acc.clone()
        m._run_fused(x, acc_t, first=False, scale=1.0 / 3.0)
        want = _bf16_emulation(m, x, pairs=pairs)
        want_acc = _bf16_emulation(m, x, acc, 1.0 / 3.0, pairs=pairs)
    assert name.startswith("resblock_bf16_kernel<"), name
    assert y.dtype == torch.bfloat16 and torch.equal(x, x0)
    for got, ref in ((y, want), (acc_t, want_acc)):
        r = ref.float().bfloat16()
        same = float((got == r).float().mean())
        rms = float(ref.pow(2).mean().sqrt())
        ulp = ref.abs().clamp_min(1e-30).log2().floor().exp2() * 2.0 ** -7            # spacing of bf16 at the reference value
        excess = float(((got.double() - ref).abs() - 0.5 * ulp).clamp_min(0).max()) / rms
        if os.environ.get("VS_TEST_VERBOSE"):
            print(f"\n   C={C} k={k} T={T} pairs={pairs}: bit-identical {same:.5f}, beyond half an ulp by at most {excess:.2e} of the rms")
        # the launch = the restated arithmetic up to the error of the fp32 accumulation chains: an operand of a later conv, a tensor
        # between two launches, or the output itself may sit on a bf16 boundary and round the other way (torch's own fp32 conv in place
        # of the fp64 one leaves 91-99 % of these outputs bit-identical; the MFMA's summation order 79-99.9 %).  Every output within half
        # a bf16 ulp of the restatement + 1.5e-2 of the rms (one such flip carried down the residual stream), the rms difference at the
        # size of the final rounding alone (2^-9 / sqrt(3) relative); the arithmetic's own error against fp64 is 3e-3 of the rms
        # (a worst-element statistic: the 14 M-element case that reaches the 512-column tiles has ten times the draws of the others -- two flips in one stream)
        assert excess <= (1.5e-2 if got.numel() < 4e6 else 3e-2), excess
        assert same >= 0.75, same
        assert float((got.double() - ref).pow(2).mean().sqrt()) <= 2e-3 * rms


def test_resblock_arithmetic_and_tensor_type_must_agree():
    """VS_MATH_SPLIT3 takes fp32 tensors and VS_MATH_BF16 bf16-resident ones: the C ABI refuses the cross combinations (no silent conversion)"""
    from visinger_amd.modules.hipconv import set_conv_math
    from visinger_amd.ops import resblock_forward
    m, _ = _block(32, 3, 1)
    ops = [c._op() for pair in zip(m.convs1, m.convs2) for c in pair]
    xb = torch.randn(1, 32, 256, device="cuda").bfloat16()
    with pytest.raises(L.VisingerHipError, match="fp32 tensors with VS_MATH_SPLIT3"):
        resblock_forward(ops, xb, torch.empty_like(xb))
    set_conv_math(m, L.MATH_BF16)
    ops = [c._op() for pair in zip(m.convs1, m.convs2) for c in pair]
    xf = torch.randn(1, 32, 256, device="cuda")
    with pytest.raises(L.VisingerHipError, match="bf16-resident tensors with VS_MATH_BF16"):
        resblock_forward(ops, xf, torch.empty_like(xf))
    with pytest.raises(L.VisingerHipError, match="acc is"):
        resblock_forward(ops, xb, torch.empty_like(xb), acc=xf)
