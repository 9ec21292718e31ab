"""conv_pipe_kernel (csrc/conv_pipe.hip): the persistent, software-pipelined instance of the 128 x 256 tile of the split-f16 x3 conv engine.
Its outputs must be BIT-IDENTICAL to conv_split_kernel<1, 8, 4, 1, 3> (same arithmetic, operation for operation: VERDICT r3 next #2) -- and
therefore stay within the engine's bounds against an fp64 convolution."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(op, x, res, acc, scale, in_act, out_act):
    from visinger_amd import _lib as L
    y = torch.empty((x.shape[0], op.rows_out, x.shape[2]), device=x.device)
    op.forward(x, y=y, res=res, acc=acc, scale=scale, in_act=in_act, out_act=out_act)
    return y, op.kernel_instance()


CASES = [
    # C_in, C_out, k, dil, B, T, res, acc, scale, lrelu, out_act
    (128, 128, 7, 1, 4, 32768, True, False, 1.0, True, 0),          # ResBlock1 conv2 (decoder.py:100-103), 128 channels
    (128, 128, 7, 5, 4, 32768, False, False, 1.0, True, 0),         # conv1, dilation 5
    (128, 128, 11, 3, 5, 26368, True, True, 1.0 / 3.0, True, 0),    # last conv of a block: + MRF accumulator, * 1/3; 515 tiles: ragged per-workgroup tile counts
    (128, 128, 3, 1, 3, 44032, True, False, 1.0, False, 0),         # 24 steps per tile: events every step
    (256, 256, 7, 3, 2, 32768, True, False, 1.0, True, 0),          # 256 channels: two row blocks, 16 chunks
    (256, 256, 3, 5, 9, 8192, True, True, 0.5, True, 2),            # short items (32 tiles per item): every other tile is an edge tile; relu
    (128, 256, 5, 1, 4, 16384, False, False, 1.0, False, 1),        # C_in != C_out, tanh
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "c%d-%d_k%d_d%d_B%d_T%d%s%s" % (c[0], c[1], c[2], c[3], c[4], c[5], "_res" if c[6] else "", "_acc" if c[7] else ""))
def test_pipe_kernel_is_bit_identical_to_the_tile_kernel(case, vs_option):
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    cin, cout, k, d, B, T, use_res, use_acc, scale, lrelu, out_act = case
    vs_option("VS_CONV_MATH", 3)
    g = torch.Generator(device="cuda").manual_seed(1000 + k * 10 + d)
    op = ConvOp(L.CONV1D, cin, cout, k, d, (k * d - d) // 2)
    w = torch.randn(cout, cin, k, device="cuda", generator=g) * (cin * k) ** -0.5
    bias = torch.randn(cout, device="cuda", generator=g) * 0.1
    op.set_weights(w, None, bias)
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    x[:, : cin // 2] *= torch.exp2(torch.randint(-6, 7, (B, cin // 2, 1), device="cuda", generator=g).float())      # per-channel scales: the running tile exponent moves
    x[0, 3, 1000:1300] = 2.0 ** 9                                                                                       # ... and rescales accumulators mid-tile
    res = torch.randn(B, cout, T, device="cuda", generator=g) if use_res else None
    acc = torch.randn(B, cout, T, device="cuda", generator=g) if use_acc else None
    in_act = L.IN_LRELU if lrelu else L.IN_NONE
    vs_option("VS_PIPE", 0)
    y_ref, k_ref = _run(op, x, res, acc, scale, in_act, out_act)
    vs_option("VS_PIPE", 1)
    y_pipe, k_pipe = _run(op, x, res, acc, scale, in_act, out_act)
    assert k_ref == "conv_split_kernel<1, 8, 4, 1, 3>" and k_pipe == "conv_pipe_kernel<%d, %s>" % (k, "true" if use_acc else "false"), (k_ref, k_pipe)
    assert torch.equal(y_pipe, y_ref), float((y_pipe - y_ref).abs().max())
    y2, _ = _run(op, x, res, acc, scale, in_act, out_act)                # and run-to-run
    assert torch.equal(y2, y_pipe)
    # against fp64 on the first 4096 columns of item 0 (left edge tile included): the engine's fp32-class bound (tests/test_conv_split_gpu.py)
    xs = x[:1, :, :8192].double()
    xs = torch.where(xs > 0, xs, 0.1 * xs) if lrelu else xs
    ref = torch.nn.functional.conv1d(xs, w.double(), bias.double(), padding=(k * d - d) // 2, dilation=d)[:, :, :4096]
    if use_res:
        ref = ref + res[:1, :, :4096].double()
    if use_acc:
        ref = ref + acc[:1, :, :4096].double()
    ref = ref * scale
    ref = torch.tanh(ref) if out_act == 1 else (torch.relu(ref) if out_act == 2 else ref)
    err = (y_pipe[:1, :, :4096].double() - ref)
    assert float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) <= 2e-6, float(err.abs().max())


def test_pipe_kernel_is_not_taken_where_its_preconditions_fail(vs_option):
    """masks, per-item bias, ragged column counts, short launches: the tile kernel (same results as before this round)"""
    from visinger_amd import _lib as L
    from visinger_amd.ops import ConvOp
    vs_option("VS_CONV_MATH", 3)
    vs_option("VS_PIPE", 1)
    op = ConvOp(L.CONV1D, 128, 128, 7, 1, 3)
    op.set_weights(torch.randn(128, 128, 7, device="cuda") * 0.03, None, torch.zeros(128, device="cuda"))
    x = torch.randn(4, 128, 32768, device="cuda")
    op.forward(x)
    assert op.kernel_instance().startswith("conv_pipe_kernel")
    mask = torch.ones(4, 32768, device="cuda")
    op.forward(x, mask=mask, out_mask=True)
    assert op.kernel_instance() == "conv_split_kernel<1, 8, 4, 1, 3>"
    op.forward(x, mask=mask, in_act=L.IN_LRELU_MASK)
    assert op.kernel_instance() == "conv_split_kernel<1, 8, 4, 1, 3>"
    op.forward(torch.randn(4, 128, 32768 + 128, device="cuda"))                  # N % 256 != 0
    assert op.kernel_instance() == "conv_split_kernel<1, 8, 4, 1, 3>"
    op.forward(torch.randn(1, 128, 65536, device="cuda"))                        # 256 tiles: fewer than two per workgroup
    assert not op.kernel_instance().startswith("conv_pipe_kernel")
    vs_option("VS_PIPE", 0)                                                      # the default: opt-in only
    op.forward(x)
    assert op.kernel_instance() == "conv_split_kernel<1, 8, 4, 1, 3>"
