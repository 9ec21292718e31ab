#!/usr/bin/env python3
"""Error of each arithmetic of the conv engine against an fp64 convolution of the same fp32 inputs:
fp32 MFMA (direct), fp32 Winograd F(2,3), split-bf16 x6, bf16, and torch's own fp32 conv on the GPU for scale.
    python tools/conv_accuracy.py
Prints RMS and max error relative to the RMS of the output."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402


def run(C, k, d, T, B=2, seed=0, scale_x=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(B, C, T, device="cuda", generator=g) * scale_x
    w = torch.randn(C, C, k, device="cuda", generator=g) / (C * k) ** 0.5
    bias = torch.randn(C, device="cuda", generator=g)
    pad = d * (k - 1) // 2
    ref = torch.nn.functional.conv1d(x.double(), w.double(), bias.double(), padding=pad, dilation=d)
    rms = ref.pow(2).mean().sqrt().item()
    out = {}
    for name, env, math in (("fp32 mfma", {"VS_NO_WINO": "1"}, L.MATH_F32), ("fp32 F(2,3)", {"VS_WINO_FORCE": "1"}, L.MATH_F32),
                            ("split-bf16 x6", {}, L.MATH_SPLIT6), ("split-f16 x3", {}, L.MATH_SPLIT3), ("bf16", {}, L.MATH_BF16)):
        for kk in ("VS_NO_WINO", "VS_WINO_FORCE"):
            L.set_option(kk, int(env.get(kk, 0)))
        op = ConvOp(L.CONV1D, C, C, k, d, pad).set_math(math)
        op.set_weights(w, None, bias)
        y = op.forward(x)
        e = (y.double() - ref)
        out[name] = (e.pow(2).mean().sqrt().item() / rms, e.abs().max().item() / rms, op.kernel_instance())
    for kk in ("VS_NO_WINO", "VS_WINO_FORCE"):
        L.set_option(kk, 0)
    y = torch.nn.functional.conv1d(x, w, bias, padding=pad, dilation=d)
    e = y.double() - ref
    out["torch fp32 (MIOpen)"] = (e.pow(2).mean().sqrt().item() / rms, e.abs().max().item() / rms, "-")
    return out


if __name__ == "__main__":
    torch.backends.cudnn.allow_tf32 = False
    for C, k, d, T in ((128, 3, 1, 4096), (128, 7, 3, 4096), (256, 11, 1, 2048), (256, 11, 5, 2048), (64, 11, 1, 8192), (192, 5, 1, 1024)):
        print(f"C={C} k={k} d={d} T={T}")
        for name, (r, m, inst) in run(C, k, d, T).items():
            print(f"   {name:22s} rms {r:.3e}   max {m:.3e}   {inst}")
