#!/usr/bin/env python3
"""Per-tile shader-clock stamps of relattn_dma_kernel (workgroup 0, wave 0): wait + barrier, DMA issue, S^T MFMAs, softmax, P V MFMAs."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import rel_attention
B, nh, dk, T = 8, 2, 256, 4096
qkv = torch.randn(B, 3 * nh * dk, T, device="cuda")
rel_k = torch.randn(1, 9, dk, device="cuda") * dk ** -0.5
rel_v = torch.randn(1, 9, dk, device="cuda") * dk ** -0.5
mask = torch.ones(B, T, device="cuda")
for _ in range(2): rel_attention(qkv, nh, rel_k, rel_v, mask, 4, math=L.MATH_BF16)
buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
lib = L.lib(); lib.vs_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.vs_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
rel_attention(qkv, nh, rel_k, rel_v, mask, 4, math=L.MATH_BF16)
torch.cuda.synchronize()
lib.vs_debug_set_stamp_buffer(None)
print(lib.vs_last_kernel_name().decode())
a = buf.cpu().numpy()[:960].reshape(120, 8)
seg = np.stack([a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 4] - a[:, 3], a[:, 5] - a[:, 4]], 1)[8:]
per = np.diff(a[8:, 0])
print("tile period: median %d mean %d cycles" % (np.median(per), per.mean()))
print("median cycles: wait+barrier %d, dma issue %d, S^T %d, softmax %d, PV %d, loop-back %d" % (*np.median(seg, 0), np.median(a[9:, 0] - a[8:-1, 5])))
print("mean   cycles: wait+barrier %d, dma issue %d, S^T %d, softmax %d, PV %d" % tuple(seg.mean(0)))
for i in range(40, 52): print("  tile", i + 8, seg[i].tolist())
