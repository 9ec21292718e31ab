#!/usr/bin/env python3
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import rel_attention
def case(dk, nh, T, ws, B, pre_f32, pre_old):
    g = torch.Generator().manual_seed(dk * 3 + T)
    C = dk * nh
    qkv = torch.randn(B, 3 * C, T, generator=g).cuda()
    rel_k = (torch.randn(1, 9, dk, generator=g) * dk ** -0.5).cuda()
    rel_v = (torch.randn(1, 9, dk, generator=g) * dk ** -0.5).cuda()
    lens = torch.tensor([T, max(1, (2 * T) // 3), 0])[:B]
    mask = (torch.arange(T)[None] < lens[:, None]).float().cuda()
    if pre_f32: rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_F32)
    if pre_old:
        L.set_option("VS_NO_ATTN_DMA", 1)
        rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=False)
        L.set_option("VS_NO_ATTN_DMA", 0)
    outs = [rel_attention(qkv, nh, rel_k, rel_v, mask, ws, math=L.MATH_BF16, ksplit_auto=False) for _ in range(4)]
    torch.cuda.synchronize()
    res = []
    for i in range(1, 4):
        d = (outs[i] - outs[0]).abs()
        nz = torch.nonzero(d > 0)
        res.append("same" if len(nz) == 0 else f"DIFF n={len(nz)} max={float(d.max()):.1e} items={torch.unique(nz[:,0]).tolist()} q=[{int(nz[:,2].min())},{int(nz[:,2].max())}]")
    print(dk, T, B, "f32" if pre_f32 else "", "old" if pre_old else "", res)
case(256, 2, 1028, 4, 3, False, False)
case(256, 2, 1028, 4, 3, True, False)
case(256, 2, 1028, 4, 3, False, True)
case(256, 2, 1028, 4, 3, True, True)
case(256, 2, 4096, 4, 2, True, True)
case(256, 2, 1028, 4, 3, True, True)
