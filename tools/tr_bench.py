#!/usr/bin/env python3
"""The generator's four upsamplers (nn.ConvTranspose1d behind a leaky-relu, decoder.py:36-48) at the headline shapes (B = 32, T_mel = 1024, hop 256): kernel instance,
microseconds per launch (HIP events, 20 launches after 5), algorithmic TFLOP/s and GB/s.  VS_NO_KTAP=1 / VS_NO_TR_EPI=1: the tile kernel / the generic epilogue.
Usage (GPU box): python tools/tr_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402

B = int(os.environ.get("TR_B", "32"))
for cin, cout, k, st, T in ((512, 256, 16, 8, 1024), (256, 128, 16, 8, 8192), (128, 64, 4, 2, 65536), (64, 32, 4, 2, 131072)):
    pad = (k - st) // 2
    op = ConvOp(L.CONV_TRANSPOSE1D, cin, cout, k, st, pad)
    op.set_weights(torch.randn(cin, cout, k, device="cuda") * (cin * k / st) ** -0.5, None, torch.randn(cout, device="cuda"))
    x = torch.randn(B, cin, T, device="cuda")
    y = torch.empty((B, cout, op.out_len(T)), device="cuda")
    for _ in range(5):
        op.forward(x, y=y, in_act=L.IN_LRELU)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        op.forward(x, y=y, in_act=L.IN_LRELU)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    fl = 2.0 * B * cin * cout * (k // st) * y.shape[2]
    by = 4.0 * (x.numel() + y.numel())
    print(f"{cin:4d} -> {cout:3d} k={k:2d} s={st} T_in={T:6d}: {op.kernel_instance():44s} {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  {by / us / 1e3:6.0f} GB/s")
