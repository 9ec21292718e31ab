#!/usr/bin/env python3
"""Debug: where conv_pipe_kernel differs from conv_split_kernel<1, 8, 4, 1, 3> (element positions of mismatches)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp
C, k, d, T, B = int(os.environ.get("C", 128)), int(os.environ.get("K", 7)), int(os.environ.get("D", 1)), int(os.environ.get("T", 32768)), int(os.environ.get("B", 4))
RES, ACC, ACT = int(os.environ.get("RES", 1)), int(os.environ.get("ACC", 0)), int(os.environ.get("ACT", 1))
L.set_option("VS_CONV_MATH", 3)
g = torch.Generator(device="cuda").manual_seed(5)
op = ConvOp(L.CONV1D, C, C, k, d, (k * d - d) // 2)
op.set_weights(torch.randn(C, C, k, device="cuda", generator=g) * (C * k) ** -0.5, None, torch.randn(C, device="cuda", generator=g) * 0.1)
x = torch.randn(B, C, T, device="cuda", generator=g)
res = torch.randn(B, C, T, device="cuda", generator=g) if RES else None
acc = torch.randn(B, C, T, device="cuda", generator=g) if ACC else None
ys = {}
for nopipe in (1, 0):
    L.set_option("VS_PIPE", 1 - nopipe)
    y = torch.full((B, C, T), 7.0, device="cuda")
    op.forward(x, y=y, res=res, acc=acc, in_act=L.IN_LRELU if ACT else L.IN_NONE)
    torch.cuda.synchronize()
    ys[nopipe] = y.cpu().numpy()
    print(nopipe, op.kernel_instance())
a, b = ys[1], ys[0]
bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
print("mismatches", bad.sum(), "of", bad.size, " nan in pipe:", np.isnan(b).sum(), " untouched (7.0):", (b == 7.0).sum())
if bad.sum():
    bi, ci, ti = np.nonzero(bad)
    print("items", np.unique(bi, return_counts=True))
    print("rows (mod 32) hist", np.bincount(ci % 32, minlength=32))
    print("row tiles", np.bincount(ci // 32))
    print("col in tile hist (by 32)", np.bincount((ti % 256) // 32, minlength=8))
    print("col in tile (mod 32) hist", np.bincount(ti % 32, minlength=32))
    tiles = ti // 256 + bi * (T // 256)
    u, cnt = np.unique(tiles, return_counts=True)
    print("tiles with mismatches:", len(u), "of", B * T // 256, " first", u[:20], cnt[:20], " last", u[-10:])
    print("max abs diff", np.nanmax(np.abs(a - b)), " rel rms", np.sqrt(np.nanmean((a - b) ** 2) / np.mean(a ** 2)))
    i = np.flatnonzero(bad.ravel())[:8]
    print("examples", [(np.unravel_index(j, bad.shape), a.ravel()[j], b.ravel()[j]) for j in i])
