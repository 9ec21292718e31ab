#!/usr/bin/env python3
"""Split-f16 x3 arithmetic (VS_MATH_SPLIT3) against an fp64 convolution: shapes of every tile family, inputs whose magnitude grows from
channel chunk to channel chunk (the tile's running scale is lowered and the accumulators follow), tiny / huge inputs, transposed and
paired kernels, residual / accumulate / mask / bias_b epilogues.  Prints error relative to the output rms next to the split-bf16 x6
engine's on the same case.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402

torch.manual_seed(0)
worst = 0.0


def case(name, Cin, Cout, k, d, T, B=2, xscale=1.0, ramp=0.0, transposed=False, res=False, acc=False, mask=False, bias_b=False, in_act=L.IN_NONE):
    global worst
    g = torch.Generator(device="cuda").manual_seed(hash(name) & 0xffff)
    x = torch.randn(B, Cin, T, device="cuda", generator=g) * xscale
    if ramp:      # channel c scaled by 10^(ramp * c / Cin): later chunks dominate -> rescale events
        x = x * (10.0 ** (ramp * torch.arange(Cin, device="cuda") / Cin))[None, :, None]
    if transposed:
        u = d
        w = torch.randn(Cin, Cout, k, device="cuda", generator=g) / (Cin * k / u) ** 0.5
        pad = (k - u) // 2
        kind = L.CONV_TRANSPOSE1D
    else:
        w = torch.randn(Cout, Cin, k, device="cuda", generator=g) / (Cin * k) ** 0.5
        pad = d * (k - 1) // 2
        kind = L.CONV1D
    bias = torch.randn(Cout, device="cuda", generator=g)
    xin = x.double()
    m2 = None
    if in_act in (L.IN_LRELU, L.IN_LRELU_MASK):
        xin = torch.nn.functional.leaky_relu(xin, 0.1)
    if mask:
        m2 = (torch.rand(B, T, device="cuda", generator=g) > 0.2).float()
        xin = xin * m2[:, None, :].double()
    if transposed:
        ref = torch.nn.functional.conv_transpose1d(xin, w.double(), bias.double(), stride=d, padding=pad)
    else:
        ref = torch.nn.functional.conv1d(xin, w.double(), bias.double(), padding=pad, dilation=d)
    bb = None
    if bias_b:
        bb = torch.randn(B, Cout, device="cuda", generator=g)
        ref = ref + bb[:, :, None].double()
    r = a = None
    if res:
        r = torch.randn(B, Cout, ref.shape[2], device="cuda", generator=g)
        ref = ref + r.double()
    if acc:
        a = torch.randn(B, Cout, ref.shape[2], device="cuda", generator=g)
        ref = ref + a.double()
    rms = ref.pow(2).mean().sqrt().item()
    out = {}
    for math in (L.MATH_SPLIT6, L.MATH_SPLIT3):
        op = ConvOp(kind, Cin, Cout, k, d, pad).set_math(math)
        op.set_weights(w, None, bias)
        ia = in_act if not mask else (L.IN_LRELU_MASK if in_act == L.IN_LRELU else L.IN_MASK)
        y = op.forward(x, in_act=ia, mask=m2, res=r, acc=a, bias_b=bb)
        e = y.double() - ref
        out[math] = (e.pow(2).mean().sqrt().item() / rms, e.abs().max().item() / rms, op.kernel_instance())
    r6, r3 = out[L.MATH_SPLIT6], out[L.MATH_SPLIT3]
    flag = "" if r3[0] <= max(2.0 * r6[0], 1e-6) else "   <-- WORSE"
    worst = max(worst, r3[0] / max(r6[0], 1e-9))
    print(f"{name:44s} split6 rms {r6[0]:.2e} max {r6[1]:.2e} | split3 rms {r3[0]:.2e} max {r3[1]:.2e}  {r3[2]}{flag}", flush=True)


case("C=128 k=3", 128, 128, 3, 1, 4096)
case("C=128 k=7 d=3 res", 128, 128, 7, 3, 4096, res=True, in_act=L.IN_LRELU)
case("C=256 k=11 res+acc", 256, 256, 11, 1, 2048, res=True, acc=True, in_act=L.IN_LRELU)
case("C=64 k=7 (64-row tile)", 64, 64, 7, 1, 8192, res=True, in_act=L.IN_LRELU)
case("C=32 k=3 (32-row tile)", 32, 32, 3, 5, 8192, res=True, in_act=L.IN_LRELU)
case("192->384 k=5", 192, 384, 5, 1, 1024, B=4)
case("192->192 k=1 mask bias_b", 192, 192, 1, 1, 1000, B=3, mask=True, bias_b=True)
case("Cin=70 (ragged chunk) T=333", 70, 96, 5, 1, 333)
case("tiny inputs x 1e-6", 128, 128, 7, 1, 2048, xscale=1e-6)
case("huge inputs x 1e6", 128, 128, 7, 1, 2048, xscale=1e6)
case("ramp 10^0..10^6 over channels", 128, 128, 3, 1, 2048, ramp=6.0)
case("ramp 10^0..10^-6 over channels", 128, 128, 3, 1, 2048, ramp=-6.0)
case("ramp 10^0..10^12, C=256 k=7", 256, 128, 7, 1, 1024, ramp=12.0)
case("transposed 256->128 k16 u8", 256, 128, 16, 8, 512, transposed=True, in_act=L.IN_LRELU)
case("transposed 64->32 k4 u2", 64, 32, 4, 2, 4096, transposed=True, in_act=L.IN_LRELU)
case("short T=5", 96, 96, 9, 1, 5)
case("single utterance B=1 T=1024 (small tiles)", 192, 192, 3, 1, 1024, B=1)
# the paired kernel (WaveNet gate): against torch in fp64
x = torch.randn(2, 192, 700, device="cuda")
w = torch.randn(384, 192, 5, device="cuda") / (192 * 5) ** 0.5
b = torch.randn(384, device="cuda")
pre = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), padding=2)
ref = torch.tanh(pre[:, :192]) * torch.sigmoid(pre[:, 192:])
for math in (L.MATH_SPLIT6, L.MATH_SPLIT3):
    op = ConvOp(L.CONV1D_PAIRED, 192, 384, 5, 1, 2).set_math(math)
    op.set_weights(w, None, b)
    y = op.forward(x, pair_mode=L.PAIR_GATE)
    e = y.double() - ref
    print(f"paired gate math {math}: rms {e.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.2e}  {op.kernel_instance()}")
print(f"worst split3 / split6 rms ratio: {worst:.2f}")
