#!/usr/bin/env python3
"""Round 5: the plain-bf16 instances of the TILE kernel (conv_split_kernel<..., 1>, csrc/conv_split_body.inc) with a masked input transform were found to
RACE at production sizes (B >= 2, T = 4096, two workgroups per CU): an all-ones mask changes the result, run-to-run results differ, always in the
lanes 32-63 / 48-63 of a staged column group.  This probe runs the legacy instances (VS_NO_KTAP=1) and the conv_ktap instances on the same data:
masked(== 1) against unmasked, and three masked runs against each other.  Usage: python tools/mask_race_probe.py   (VS_LIB=build/<variant>/... for A/B libraries)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L                     # noqa: E402
from visinger_amd.ops import ConvOp                    # noqa: E402

L.set_option("VS_NO_SMALL_GRID", 1)


def go(noktap, math, cfg, cin, cout, k, B, T):
    L.set_option("VS_CONV_MATH", math)
    L.set_option("VS_CONV_CFG", cfg)
    L.set_option("VS_NO_KTAP", noktap)
    g = torch.Generator(device="cuda").manual_seed(1)
    op = ConvOp(L.CONV1D, cin, cout, k, 1, k // 2)
    w = torch.randn(cout, cin, k, device="cuda", generator=g) * (cin * k) ** -0.5
    op.set_weights(w, None, torch.zeros(cout, device="cuda"))
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    mask = torch.ones(B, T, device="cuda")
    a = op.forward(x, in_act=L.IN_NONE).clone()
    outs = [op.forward(x, in_act=L.IN_MASK, mask=mask).clone() for _ in range(3)]
    name = op.kernel_instance()
    d = (a - outs[0]).abs()
    bad = d > 0
    s = (f"math {math} cfg {cfg} {cin}->{cout} k{k} B{B} T{T} {name}: masked(==1) vs unmasked max {float(d.max()):.3g} nbad {int(bad.sum())}; "
         f"run-to-run equal {all(torch.equal(outs[0], o) for o in outs[1:])}")
    if bad.any():
        idx = bad.nonzero()
        s += f" col/16%16 {sorted(set(((idx[:, 2] // 16) % 16).tolist()))}"
    print(s, flush=True)


for noktap in (1, 0):
    for math in (1, 3):
        for cfg in (0, 3):
            for k in (1, 9):
                go(noktap, math, cfg, 512, 1536, k, 2, 4096)
    go(noktap, 1, 0, 512, 1536, 1, 8, 4096)
