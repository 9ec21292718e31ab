#!/usr/bin/env python3
"""Micro-benchmark of vs_relattn_fwd at the production shape (B=32, 2 heads x 96, T=1024)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd.ops import rel_attention
B, C, T, nh = 32, 192, int(os.environ.get("T", 1024)), 2
qkv = torch.randn(B, 3 * C, T, device="cuda")
rk = torch.randn(1, 9, C // nh, device="cuda") * 0.1
rv = torch.randn(1, 9, C // nh, device="cuda") * 0.1
mask = torch.ones(B, T, device="cuda")
out = torch.empty(B, C, T, device="cuda")
for _ in range(2):
    rel_attention(qkv, nh, rk, rv, mask, 4, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    rel_attention(qkv, nh, rk, rv, mask, 4, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 4.0 * B * nh * T * T * (C // nh)
print(f"relattn B={B} T={T}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s (QK^T + PV only)")
