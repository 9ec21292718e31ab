#!/usr/bin/env python3
"""Timing-only perturbations of relattn_dma_kernel (library built with -DVS_ATTN_PERTURB, through VS_LIB)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import rel_attention
B, nh, dk, T = 8, 2, 256, 4096
qkv = torch.randn(B, 3 * nh * dk, T, device="cuda")
rel_k = torch.randn(1, 9, dk, device="cuda") * dk ** -0.5
rel_v = torch.randn(1, 9, dk, device="cuda") * dk ** -0.5
mask = torch.ones(B, T, device="cuda")
def t():
    for _ in range(2): rel_attention(qkv, nh, rel_k, rel_v, mask, 4, math=L.MATH_BF16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): rel_attention(qkv, nh, rel_k, rel_v, mask, 4, math=L.MATH_BF16)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
for dbg in (0, 1, 2, 3, 4, 7, 8, 16, 32, 48, 63):
    L.set_option("VS_SPLIT_DBG", dbg)
    print(f"dbg {dbg:3d}: {t():8.1f} us (incl. ~30 us of K / V packing)", flush=True)
