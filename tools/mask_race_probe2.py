#!/usr/bin/env python3
"""Round 6: WHAT the masked plain-bf16 tile kernel stages wrongly.  A 512 -> 512 1 x 1 conv with the IDENTITY as its weight returns bf16(x) exactly
(one non-zero product per output, exact in fp32): every output that differs from bf16(x) names the staged element (item, channel, frame) that was wrong and shows
the value that took its place.  Usage: python tools/mask_race_probe2.py [cfg]   (legacy instances: VS_NO_KTAP=1 is set here)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L                     # noqa: E402
from visinger_amd.ops import ConvOp                    # noqa: E402

L.set_option("VS_NO_SMALL_GRID", 1)
L.set_option("VS_NO_KTAP", 1)
L.set_option("VS_CONV_MATH", 1)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.set_option("VS_CONV_CFG", cfg)
C, B, T = 512, 8, 4096      # (16 x 4 x 8 = 512 workgroups of 128 x 256: two per CU -- with one per CU the kernel is clean)
g = torch.Generator(device="cuda").manual_seed(1)
op = ConvOp(L.CONV1D, C, C, 1, 1, 0)
op.set_weights(torch.eye(C, device="cuda").reshape(C, C, 1).contiguous(), None, torch.zeros(C, device="cuda"))
x = torch.randn(B, C, T, device="cuda", generator=g)
want = x.to(torch.bfloat16).float()
mask = torch.ones(B, T, device="cuda")
for rep in range(3):
    y = op.forward(x, in_act=L.IN_MASK, mask=mask).clone()
    bad = (y != want)
    idx = bad.nonzero()
    print(f"rep {rep} {op.kernel_instance()}: {int(bad.sum())} wrong of {y.numel()}", flush=True)
    if not len(idx):
        continue
    b, c, t = idx[:, 0], idx[:, 1], idx[:, 2]
    print("  channel % 16:", sorted(set((c % 16).tolist())), " chunk (channel // 16):", sorted(set((c // 16).tolist()))[:40])
    print("  t % 256 // 16:", sorted(set(((t % 256) // 16).tolist())), " items:", sorted(set(b.tolist())))
    got = y[bad]
    print("  wrong values that are exactly 0:", int((got == 0).sum()), " of", len(got))
    # is the wrong value the staged value of another chunk (same lane, same channel-in-chunk)?  try chunk offsets -4 .. 4
    for dc in (-4, -3, -2, -1, 1, 2, 3, 4):
        c2 = c + 16 * dc
        ok = (c2 >= 0) & (c2 < C)
        hit = torch.zeros_like(ok)
        hit[ok] = want[b[ok], c2[ok], t[ok]] == got[ok]
        if int(hit.sum()):
            print(f"  equals bf16(x) of chunk {dc:+d} (same channel-in-chunk, same frame): {int(hit.sum())}")
    for dt in (-256, -128, -64, 64, 128, 256):
        t2 = t + dt
        ok = (t2 >= 0) & (t2 < T)
        hit = torch.zeros_like(ok)
        hit[ok] = want[b[ok], c[ok], t2[ok]] == got[ok]
        if int(hit.sum()):
            print(f"  equals bf16(x) at frame {dt:+d}: {int(hit.sum())}")
    for dj in (-3, -2, -1, 1, 2, 3):
        c2 = c + dj
        ok = (c2 >= 0) & (c2 < C) & ((c2 // 4) == (c // 4))
        hit = torch.zeros_like(ok)
        hit[ok] = want[b[ok], c2[ok], t[ok]] == got[ok]
        if int(hit.sum()):
            print(f"  equals bf16(x) of channel {dj:+d} (same wave's four): {int(hit.sum())}")
    k = min(12, len(idx))
    for i in range(k):
        print(f"    b{int(b[i])} c{int(c[i])} t{int(t[i])}: got {float(got[i]):+.6f} want {float(want[b[i], c[i], t[i]]):+.6f} x {float(x[b[i], c[i], t[i]]):+.6f}")
