"""relattn_dma_kernel (csrc/attention_dma.hip, round 4): the plain-bf16 attention of wide heads (129 .. 256 channels, BASELINE config 5) with the
pre-packed K / V tile images brought into an LDS ring by LDS-DMA.  Same algorithm and arithmetic as relattn_bf16_kernel<DT, 32, 1, true>
except for the summation order of the score tile (even / odd k-steps in two accumulators): held to the same stated bf16 bounds against the
exact-fp32 kernel (which the reference's golden vectors and the fp64 oracle pin), and to a much tighter bound against the kernel it
replaces -- the two round the same bf16 operands, so they differ by fp32 summation order and by the bf16 probabilities that flip on it."""
import pytest
import torch

from visinger_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dk,nh,T,ws,share,B", [(256, 2, 4096, 4, True, 2), (256, 2, 1028, 4, True, 3), (192, 2, 2048, 4, True, 2), (160, 1, 1056, 7, False, 2),
                                                (256, 1, 1024, None, True, 2), (224, 2, 1536, 1, True, 1)])
def test_dma_attention_matches_the_fp32_kernel_and_the_kernel_it_replaces(dk, nh, T, ws, share, B, vs_option):
    from visinger_amd.ops import rel_attention
    g = torch.Generator().manual_seed(dk * 3 + T)
    C = dk * nh
    qkv = torch.randn(B, 3 * C, T, generator=g).cuda()
    nrel = 0 if ws is None else 2 * ws + 1
    rel_k = (torch.randn(1 if share else nh, nrel, dk, generator=g) * dk ** -0.5).cuda() if nrel else None
    rel_v = (torch.randn(1 if share else nh, nrel, dk, generator=g) * dk ** -0.5).cuda() if nrel else None
    lens = torch.tensor([T, max(1, (2 * T) // 3), 0])[:B]
    mask = (torch.arange(T)[None] < lens[:, None]).float().cuda()
    ref = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_F32)
    vs_option("VS_NO_ATTN_DMA", 1)
    old = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=False)
    assert L.lib().vs_last_kernel_name().decode().startswith("relattn_bf16_kernel<")
    vs_option("VS_NO_ATTN_DMA", 0)
    vs_option("VS_ATTN_DMA_ONE_WAVE", 1)
    one = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=False)        # the round-4 form: one wave per query group, all output tiles
    assert L.lib().vs_last_kernel_name().decode() == "relattn_dma_kernel<%d, 1>" % (6 if dk <= 192 else 8)
    vs_option("VS_ATTN_DMA_ONE_WAVE", 0)
    got = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=False)        # round 6: a wave pair splits the head's channels
    assert L.lib().vs_last_kernel_name().decode() == "relattn_dma_kernel<%d, 2>" % (6 if dk <= 192 else 8)
    # (the pair sums the score tile as (first half of the channels) + (second half) where the single wave sums even + odd k-steps: another fp32 order, the same
    #  bf16 operands -- held to the bound between the two older kernels)
    d1 = (got - one).abs()
    assert float(d1.pow(2).mean().sqrt()) <= 1e-3 * float(ref.pow(2).mean().sqrt()) and float(d1.max()) <= 2e-2 * max(float(ref.pow(2).mean().sqrt()), 1e-3)
    assert torch.isfinite(got).all()
    scale = float(ref.pow(2).mean().sqrt())
    err = (got - ref).abs()
    assert float(err.pow(2).mean().sqrt()) <= 1e-2 * scale and float(err.max()) <= 0.1 * max(scale, 1e-3)          # the stated bf16 bound
    d = (got - old).abs()
    print(f"dma attention dk={dk} T={T}: vs fp32 rms {float(err.pow(2).mean().sqrt()):.2e}, vs the register-staged kernel rms {float(d.pow(2).mean().sqrt()):.2e} "
          f"max {float(d.max()):.2e}, output rms {scale:.2e}")
    assert float(d.pow(2).mean().sqrt()) <= 1e-3 * scale and float(d.max()) <= 2e-2 * max(scale, 1e-3)
    again = rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=False)                                       # no atomics, no races on the ring
    assert torch.equal(again, got), float((again - got).abs().max())
