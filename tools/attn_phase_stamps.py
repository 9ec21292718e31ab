#!/usr/bin/env python3
"""Where a launch of relattn_bf16_kernel spends its time outside the tile loop (workgroup (0, 0, 0), wave 0; a -DVS_ATTN_STAMPS build:
python tools/build_variant.py attnstamps -DVS_ATTN_STAMPS attention_bf16.hip; VS_LIB=build/attnstamps/libvisinger_hip.so)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import rel_attention
WIDE = "--wide" in sys.argv      # relattn_dma_kernel (BASELINE configs[4]'s heads, plain bf16): its phase stamps are compiled in (slots 1000 ..)
B, nh, dk, T = (8, 2, 256, 4096) if WIDE else (32, 2, 96, 1024)
math = L.MATH_BF16 if WIDE else (L.MATH_SPLIT6 if "--split6" in sys.argv else L.MATH_SPLIT3)
qkv = torch.randn(B, 3 * nh * dk, T, device="cuda")
rel_k = torch.randn(1, 9, dk, device="cuda") * dk ** -0.5
rel_v = torch.randn(1, 9, dk, device="cuda") * dk ** -0.5
mask = torch.ones(B, T, device="cuda")
for _ in range(3): rel_attention(qkv, nh, rel_k, rel_v, mask, 4, math=math)
buf = torch.zeros(4096, dtype=torch.int64, device="cuda")
lib = L.lib(); lib.vs_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.vs_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
rel_attention(qkv, nh, rel_k, rel_v, mask, 4, math=math)
e1.record()
torch.cuda.synchronize()
lib.vs_debug_set_stamp_buffer(None)
print(lib.vs_last_kernel_name().decode(), "event interval %.1f us" % (e0.elapsed_time(e1) * 1e3))
a = buf.cpu().numpy()
if WIDE:
    w = a[1000:1007]
    tot = w[6] - w[0]
    for i, n in enumerate(["rel-key table to LDS", "query fragments + logits (+ pair exchange)", "first DMA pieces", "tile loop", "finish (normalise, rel-value)", "output stores"]):
        print("  %-32s %8d cycles  %5.1f %%" % (n, w[i + 1] - w[i], 100.0 * (w[i + 1] - w[i]) / tot))
    print("  total %d cycles" % tot)
    sys.exit(0)
names = ["rel-key table, query fragments + logits", "logits to LDS", "first tile staged", "tile loop", "finish (normalise, rel-value)", "output stores"]
tot = a[6] - a[0]
for i, n in enumerate(names):
    print("  %-32s %8d ticks  %5.1f %%" % (n, a[i + 1] - a[i], 100.0 * (a[i + 1] - a[i]) / tot))
print("  total %d ticks (s_memtime: 100 MHz ticks -> %.1f us)" % (tot, tot / 100.0))
per = np.diff(a[16:16 + T // 32])
print("  tile period: median %d, first five %s" % (np.median(per), per[:5].tolist()))
t = a[128:128 + 8 * (T // 32)].reshape(-1, 8)[2:-1]
seg = np.diff(t, axis=1)
print("  inside a tile (median cycles): S^T MFMAs %d | scores, softmax, P planes %d | P V MFMAs %d | tile exponents %d | barrier %d | convert + store next tile %d | barrier %d"
      % tuple(np.median(seg, 0)))
print("  next-tile loads issued at the top of the loop: %d" % np.median(t[:, 0] - a[16 + 2:16 + 2 + len(t)]))
