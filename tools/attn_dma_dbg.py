#!/usr/bin/env python3
"""Debug: run-to-run differences of relattn_dma_kernel (where do they sit?)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import rel_attention
dk, nh, T, ws, B = int(os.environ.get("DK", 256)), 2, int(os.environ.get("T", 1028)), 4, int(os.environ.get("B", 3))
g = torch.Generator().manual_seed(dk * 3 + T)
C = dk * nh
qkv = torch.randn(B, 3 * C, T, generator=g).cuda()
rel_k = (torch.randn(1, 9, dk, generator=g) * dk ** -0.5).cuda()
rel_v = (torch.randn(1, 9, dk, generator=g) * dk ** -0.5).cuda()
lens = torch.tensor([T, max(1, (2 * T) // 3), 0])[:B]
mode = os.environ.get("MASK", "ragged")
mask = (torch.arange(T)[None] < lens[:, None]).float().cuda() if mode == "ragged" else torch.ones(B, T).cuda()
outs = [rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=False).clone() for _ in range(6)]
print(L.lib().vs_last_kernel_name().decode(), "T", T, "B", B, "mask", mode)
for i in range(1, 6):
    d = (outs[i] - outs[0]).abs()
    nz = torch.nonzero(d > 0)
    if len(nz) == 0:
        print(i, "identical"); continue
    print(i, "differs: count", len(nz), "max", float(d.max()), "items", torch.unique(nz[:, 0]).tolist(), "heads", torch.unique(nz[:, 1] // dk).tolist(),
          "queries min/max", int(nz[:, 2].min()), int(nz[:, 2].max()), "query blocks(128)", torch.unique(nz[:, 2] // 128).tolist()[:12], "waves(32)", torch.unique((nz[:, 2] % 128) // 32).tolist())
