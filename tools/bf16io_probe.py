#!/usr/bin/env python3
"""conv_split_kernel_bf16io<1, 8, 4, 1, 1, 3> (plain-bf16 arithmetic over bf16-resident tensors: BASELINE config 5's dominant instance) at the config-5
generator shapes; with VS_LIB=build/perturb/libvisinger_hip.so and VS_SPLIT_DBG=1|2|4|8 the timing-only perturbations of the main loop.
Usage (GPU box): [VS_LIB=...] [VS_SPLIT_DBG=n] python tools/bf16io_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402

L.set_option("VS_CONV_MATH", 1)
B = 8
for C, T, k in ((128, 262144, 7), (128, 262144, 11), (256, 32768, 7), (256, 32768, 3)):
    x = torch.randn(B, C, T, device="cuda").bfloat16()
    y = torch.empty_like(x)
    res = torch.randn(B, C, T, device="cuda").bfloat16()
    op = ConvOp(L.CONV1D, C, C, k, 1, (k - 1) // 2)
    op.set_weights(torch.randn(C, C, k, device="cuda") * 0.03, None, torch.randn(C, device="cuda"))
    for _ in range(2):
        op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    fl = 2.0 * C * C * k * B * T
    print(f"dbg={os.environ.get('VS_SPLIT_DBG', '0'):>2s} C={C} T={T} k={k:2d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {op.kernel_instance()}", flush=True)
