#!/usr/bin/env python3
"""Per-step shader-clock stamps of conv_pipe_kernel (perturb build through VS_LIB): where a wave's step goes."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp
C, k, T, B = int(os.environ.get("C", 128)), int(os.environ.get("K", 7)), int(os.environ.get("T", 65536)), int(os.environ.get("B", 32))
RES = int(os.environ.get("RES", 1))
L.set_option("VS_CONV_MATH", 3)
L.set_option("VS_SPLIT_DBG", int(os.environ.get("DBG", 0)))
op = ConvOp(L.CONV1D, C, C, k, 1, (k - 1) // 2)
op.set_weights(torch.randn(C, C, k, device="cuda") * 0.03, None, torch.randn(C, device="cuda"))
x = torch.randn(B, C, T, device="cuda"); y = torch.empty_like(x); res = torch.randn_like(x) if RES else None
for _ in range(2):
    op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
lib = L.lib()
lib.vs_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.vs_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
torch.cuda.synchronize()
lib.vs_debug_set_stamp_buffer(None)
print(op.kernel_instance(), "C", C, "k", k, "res", RES)
a = buf.cpu().numpy()[:2048].reshape(2, 128, 8)[:, :100]
names = ["start", "issued(stage/A/B)", "A arrived", "6 mfma issued", "tick done", "12 mfma issued", "barrier"]
for w in range(2):
    s = a[w]
    t0 = s[0, 0]
    print(f"wave {4*w}: step period (start to start) median {np.median(np.diff(s[:, 0])):.0f} cycles; mean {np.mean(np.diff(s[8:, 0])):.0f}")
    seg = np.stack([s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4]], 1)
    print("   median cycles per segment: start->issued %d, ->A arrived %d, ->6 mfma %d, tick %d, ->12 mfma %d" % tuple(np.median(seg[8:], 0)))
    print("   mean   cycles per segment: start->issued %d, ->A arrived %d, ->6 mfma %d, tick %d, ->12 mfma %d" % tuple(np.mean(seg[8:], 0)))
    rows = []
    for i in range(14, 14 + 2 * k + 2):
        rows.append(f"     step {i:3d} (tap {i % k}): " + " ".join(f"{int(v):5d}" for v in seg[i]) + (f"  barrier {int(s[i,6]-s[i,5])}" if s[i, 6] > s[i, 5] else "") + f"   next start +{int(s[i+1,0]-s[i,5])}")
    print("\n".join(rows))
