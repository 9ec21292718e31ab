#!/usr/bin/env python3
"""Debug: per-workgroup phase timeline of one fused residual-pair launch (respair_split_kernel; vs_debug_set_stamp_buffer)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp, respair_forward

C, k, d = int(os.environ.get("C", 32)), int(os.environ.get("K", 7)), int(os.environ.get("D", 1))
T, B = int(os.environ.get("T", 262144 if C == 32 else 131072)), int(os.environ.get("B", 32))
op1 = ConvOp(L.CONV1D, C, C, k, d, d * (k - 1) // 2)
op2 = ConvOp(L.CONV1D, C, C, k, 1, (k - 1) // 2)
for op in (op1, op2):
    op.set_weights(torch.randn(C, C, k, device="cuda") * 0.05, None, torch.randn(C, device="cuda"))
x = torch.randn(B, C, T, device="cuda"); y = torch.empty_like(x)
for _ in range(2):
    respair_forward(op1, op2, x, y, res=x)
buf = torch.zeros(65536 * 64, dtype=torch.int64, device="cuda")
lib = L.lib()
lib.vs_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.vs_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
respair_forward(op1, op2, x, y, res=x)
torch.cuda.synchronize()
lib.vs_debug_set_stamp_buffer(None)
s = buf.cpu().numpy().reshape(-1, 64)
s = s[s[:, 0] != 0]
t = lambda a, b: (s[:, a] - s[:, b]) / 100.0
print("workgroups", len(s), "launch span %.1f us" % ((s[:, 3].max() - s[:, 0].min()) / 100.0))
for name, v in (("prologue", t(1, 0)), ("phase 1", t(2, 1)), ("transform", t(8, 2)), ("phase 2", t(9, 8)), ("epilogue", t(3, 9)), ("total", t(3, 0))):
    print(f"{name:9s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f} us")
