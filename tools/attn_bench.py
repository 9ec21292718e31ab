#!/usr/bin/env python3
"""Micro-benchmark of vs_relattn_fwd: the production shape (B=32, 2 heads x 96, T=1024) and BASELINE config 5 (B=8, 2 heads x 256,
T=4096): exact-fp32 kernel, the bf16 kernel (math = VS_MATH_BF16) and the two split arithmetics (split-bf16 x6, split-f16 x3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import rel_attention
SHAPES = ((32, 192, 1024, 2), (8, 512, 4096, 2), (8, 192, 4096, 2))
MATHS = ((L.MATH_F32, "fp32"), (L.MATH_BF16, "bf16"), (L.MATH_SPLIT6, "split-bf16 x6"), (L.MATH_SPLIT3, "split-f16 x3"))
if "--split" in sys.argv:      # the headline's shape in the two split arithmetics only (under rocprofv3 --kernel-trace --stats: kernel durations without the host's gaps)
    SHAPES, MATHS = SHAPES[:1] + SHAPES[2:], MATHS[2:]
for B, C, T, nh in SHAPES:
    qkv = torch.randn(B, 3 * C, T, device="cuda")
    rk = torch.randn(1, 9, C // nh, device="cuda") * 0.1
    rv = torch.randn(1, 9, C // nh, device="cuda") * 0.1
    mask = torch.ones(B, T, device="cuda")
    out = torch.empty(B, C, T, device="cuda")
    for math, name in MATHS:
        for _ in range(2):
            rel_attention(qkv, nh, rk, rv, mask, 4, out=out, math=math)
        kern = L.lib().vs_last_kernel_name().decode()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            rel_attention(qkv, nh, rk, rv, mask, 4, out=out, math=math)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 4.0 * B * nh * T * T * (C // nh)
        print(f"relattn {name} B={B} C={C} T={T} [{kern}]: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s (QK^T + PV only)", flush=True)
