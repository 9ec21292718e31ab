#!/usr/bin/env python3
"""Randomised sweep of the training-side convolution gradients (engine grad-input, vs_conv_wgrad, phase-stacked strided /
transposed forms, grouped VALU kernels) against aten::convolution_backward.   python tools/backward_fuzz.py [n] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd.autograd import conv_backward, disc_conv1d  # noqa: E402
from visinger_amd.modules.hipconv import HipConv1d, HipConvTranspose1d  # noqa: E402


def rel(a, b):
    return float((a - b).abs().max()) / (1e-6 + float(b.abs().max()))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.manual_seed(int(rng.integers(1 << 30)))
    worst = {}
    for case in range(n):
        kind = ["conv", "tconv", "strided", "grouped"][case % 4]
        B = int(rng.integers(1, 4))
        if kind == "conv":
            Cin, Cout = int(rng.choice([1, 8, 24, 32, 64, 100])), int(rng.choice([2, 16, 32, 48, 96]))
            K = int(rng.choice([1, 3, 5, 7, 11])); d = int(rng.choice([1, 3, 5])) if K > 1 else 1
            T = int(rng.choice([1, 5, 50, 130, 257, 600]))
            m = HipConv1d(Cin, Cout, K, dilation=d, padding=d * (K - 1) // 2).cuda()
            w = torch.randn(Cout, Cin, K, device="cuda") / (Cin * K) ** 0.5
            x = torch.randn(B, Cin, T, device="cuda"); gy = torch.randn(B, Cout, T, device="cuda")
            gx, gw = conv_backward(m, x, w, gy, True, True)
            rx, rw, _ = torch.ops.aten.convolution_backward(gy, x, w, None, [1], [m.padding[0]], [d], False, [0], 1, [True, True, False])
            desc = f"conv B{B} {Cin}->{Cout} k{K} d{d} T{T}"
        elif kind == "tconv":
            u = int(rng.choice([2, 3, 4, 5, 8])); K = u + 2 * int(rng.integers(0, 3))
            Cin, Cout, T = int(rng.choice([8, 24, 64])), int(rng.choice([4, 12, 32])), int(rng.choice([1, 9, 40, 133]))
            m = HipConvTranspose1d(Cin, Cout, K, u, padding=(K - u) // 2).cuda()
            w = torch.randn(Cin, Cout, K, device="cuda") / (Cin * K) ** 0.5
            x = torch.randn(B, Cin, T, device="cuda")
            gy = torch.randn(B, Cout, (T - 1) * u - 2 * m.padding[0] + K, device="cuda")
            gx, gw = conv_backward(m, x, w, gy, True, True)
            rx, rw, _ = torch.ops.aten.convolution_backward(gy, x, w, None, [u], [m.padding[0]], [1], True, [0], 1, [True, True, False])
            desc = f"tconv B{B} {Cin}->{Cout} k{K} u{u} T{T}"
        else:
            groups = 1 if kind == "strided" else int(rng.choice([2, 4, 8]))
            cig = int(rng.choice([1, 4, 8])) if kind == "grouped" else int(rng.choice([1, 3, 32, 64]))
            cog = int(rng.choice([1, 4, 16])) if kind == "grouped" else int(rng.choice([1, 16, 40]))
            Cin, Cout = cig * groups, cog * groups
            s_ = int(rng.choice([1, 2, 3, 4])); K = int(rng.choice([3, 5, 15, 41])); pad = int(rng.integers(0, K // 2 + 1))
            T = int(rng.choice([K + 3, 60, 211, 700]))
            holder = torch.nn.Module()
            x = torch.randn(B, Cin, T, device="cuda", requires_grad=True)
            w = (torch.randn(Cout, cig, K, device="cuda") / (cig * K) ** 0.5).requires_grad_(True)
            b = torch.randn(Cout, device="cuda", requires_grad=True)
            y = disc_conv1d(holder, x, w, b, s_, pad, groups)
            yr = torch.nn.functional.conv1d(x, w, b, s_, pad, 1, groups)
            gy = torch.randn_like(yr)
            gx, gw, gb = torch.autograd.grad(y, [x, w, b], gy)
            rx, rw, rb = torch.autograd.grad(yr, [x, w, b], gy)
            assert rel(y.detach(), yr.detach()) <= 2e-5 and rel(gb, rb) <= 2e-5
            desc = f"{kind} B{B} {Cin}->{Cout} g{groups} k{K} s{s_} p{pad} T{T}"
        e = max(rel(gx, rx), rel(gw, rw))
        if e > worst.get(kind, (0, ""))[0]:
            worst[kind] = (e, desc)
        if not (e <= 1e-4):
            print("FAIL", desc, e)
            sys.exit(1)
    for k, (e, d) in worst.items():
        print(f"{k:8s} worst rel err {e:.2e}  ({d})")
    print(f"OK {n} cases")


if __name__ == "__main__":
    main()
