#!/usr/bin/env python3
"""Timing-only perturbations of conv_pipe_kernel (library built with -DVS_PIPE_PERTURB, loaded through VS_LIB): which part of a step costs what."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp
B = 32
L.set_option("VS_CONV_MATH", 3)
def t(op, x, y, res):
    for _ in range(2): op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4 * 1e3
for C, T, k in ((128, 65536, 7), (128, 65536, 3), (256, 8192, 7)):
    x = torch.randn(B, C, T, device="cuda"); y = torch.empty_like(x); res = torch.randn_like(x)
    op = ConvOp(L.CONV1D, C, C, k, 1, (k - 1) // 2)
    op.set_weights(torch.randn(C, C, k, device="cuda") * 0.03, None, torch.randn(C, device="cuda"))
    for use_res in (0, 1):
        L.set_option("VS_PIPE", 0); L.set_option("VS_SPLIT_DBG", 0)
        base = t(op, x, y, res if use_res else None)
        L.set_option("VS_PIPE", 1)
        row = []
        for dbg in (0, 1, 2, 3, 23, 279):
            L.set_option("VS_SPLIT_DBG", dbg)
            row.append((dbg, t(op, x, y, res if use_res else None)))
        L.set_option("VS_SPLIT_DBG", 0)
        print(f"C={C} k={k} res={use_res}: tile kernel {base:7.1f} us | pipe " + "  ".join(f"dbg{d}:{v:7.1f}" for d, v in row), flush=True)
