#!/usr/bin/env python3
"""Experiment: start every other dispatch round of conv_wsplit_kernel workgroups late (VS_WSPLIT_STAGGER x 64 cycles) so that the two
workgroups of a CU alternate their transform and MFMA phases instead of running them together."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp
L.set_option("VS_WSPLIT_FORCE", 1)
B = 32
for C, T in ((128, 65536), (256, 8192)):
    for k, d in ((3, 1), (7, 1), (11, 1), (11, 5)):
        op = ConvOp(L.CONV1D, C, C, k, d, d * (k - 1) // 2)
        op.set_weights(torch.randn(C, C, k, device="cuda") * 0.05, None, torch.randn(C, device="cuda"))
        x = torch.randn(B, C, T, device="cuda"); y = torch.empty_like(x); res = torch.randn_like(x)
        line = f"C={C} k={k} d={d}:"
        for stg in (0, 32, 64, 128, 0):
            L.set_option("VS_WSPLIT_STAGGER", int(stg))
            best = 1e9
            for rnd in range(3):
                op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 4 * 1e3)
            line += f"  stagger {stg}: {best:7.1f} us"
        print(line, flush=True)
