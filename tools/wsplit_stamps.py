#!/usr/bin/env python3
"""Debug: per-workgroup phase timeline of one conv_wsplit_kernel launch (vs_debug_set_stamp_buffer hook; s_memrealtime, 100 MHz).
Slots: 0 start, 1 first chunk staged, 8 compute of chunk 1 done, 9 chunk 2 transformed and written, 10 barrier passed, 2 main loop
done, 3 end."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp

L.set_option("VS_WSPLIT_FORCE", 1)
C, k, d, T, B = int(os.environ.get("C", 128)), int(os.environ.get("K", 3)), int(os.environ.get("D", 1)), int(os.environ.get("T", 65536)), int(os.environ.get("B", 32))
op = ConvOp(L.CONV1D, C, C, k, d, (k * d - d) // 2)
op.set_weights(torch.randn(C, C, k, device="cuda") * 0.05, None, torch.randn(C, device="cuda"))
x = torch.randn(B, C, T, device="cuda"); y = torch.empty_like(x); res = torch.randn_like(x)
for _ in range(2):
    op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
buf = torch.zeros(65536 * 64, dtype=torch.int64, device="cuda")
lib = L.lib()
lib.vs_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.vs_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
torch.cuda.synchronize()
lib.vs_debug_set_stamp_buffer(None)
print("kernel:", op.kernel_instance())
full = buf.cpu().numpy().reshape(-1, 64)
full = full[full[:, 0] != 0]
t0 = full[:, 0].min()
us = lambda a: a / 100.0
print("workgroups", len(full), "launch span %.1f us" % us(full[:, 3].max() - t0))
nch = -(-C // 16)
parts = [("prologue (first chunk staged)", full[:, 1] - full[:, 0]), ("chunk 1: compute", None), ("chunk 2: transform + write", full[:, 9] - full[:, 8]),
         ("barrier after the write", full[:, 10] - full[:, 9]), ("main loop total", full[:, 2] - full[:, 1]), ("epilogue", full[:, 3] - full[:, 2]),
         ("workgroup total", full[:, 3] - full[:, 0])]
for name, v in parts:
    if v is None:
        continue
    v = us(v.astype(np.float64))
    print(f"{name:32s} mean {v.mean():8.2f}  p10 {np.percentile(v,10):8.2f}  p50 {np.percentile(v,50):8.2f}  p90 {np.percentile(v,90):8.2f} us")
ml = us((full[:, 2] - full[:, 1]).astype(np.float64)).mean()
st = us((full[:, 10] - full[:, 8]).astype(np.float64)).mean()
print(f"per chunk: {ml / nch:.2f} us of which staging (barrier + transform + barrier) {st:.2f} us -> compute {ml / nch - st * (nch - 1) / nch:.2f} us; "
      f"MFMAs per wave and chunk: {16 * -(-k // 3) * 3} x 32 cycles")
if (full[:, 4] != 0).all() and (full[:, 5] != 0).all():
    clk = (full[:, 5] - full[:, 4]).astype(np.float64) / (full[:, 2] - full[:, 1]).astype(np.float64) * 0.1
    print(f"shader clock in the main loop: median {np.median(clk):.2f} GHz")
