#!/usr/bin/env python3
"""Fused resblock pair (csrc/resblock_pair.hip) vs the two separate launches, production shapes of the 32- / 64-channel stages."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp, respair_forward
B = 32
def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for C, T in ((32, 262144), (64, 131072)):
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            op1 = ConvOp(L.CONV1D, C, C, k, d, d * (k - 1) // 2); op2 = ConvOp(L.CONV1D, C, C, k, 1, (k - 1) // 2)
            for op in (op1, op2): op.set_weights(torch.randn(C, C, k, device="cuda") * 0.05, None, torch.randn(C, device="cuda"))
            x = torch.randn(B, C, T, device="cuda"); tmp = torch.empty_like(x); y = torch.empty_like(x)
            sep = t(lambda: (op1.forward(x, in_act=L.IN_LRELU, y=tmp), op2.forward(tmp, in_act=L.IN_LRELU, res=x, y=y)))
            names = f"{op1.kernel_instance()} + {op2.kernel_instance()}"
            fus = t(lambda: respair_forward(op1, op2, x, y, res=x))
            fl = op1.algorithmic_flops(B, T) * 2
            print(f"C={C} k={k} d={d}: separate {sep:8.1f} us ({names})   fused {fus:8.1f} us  "
                  f"{fl/fus/1e6:6.1f} TFLOP/s   x{sep/fus:.2f}", flush=True)
