#!/usr/bin/env python3
"""Debug: where does conv_wsplit_kernel differ from the direct split kernel (rows / columns of the wrong elements)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp
L.set_option("VS_WSPLIT_FORCE", 1)
Cin, Cout, k, d, T, B = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 64, 11, 3, 700, 1))]
torch.manual_seed(0)
op = ConvOp(L.CONV1D, Cin, Cout, k, d, d * (k - 1) // 2)
op.set_weights(torch.randn(Cout, Cin, k, device="cuda") * 0.05, None, torch.randn(Cout, device="cuda"))
x = torch.randn(B, Cin, T, device="cuda")
y = op.forward(x, in_act=L.IN_LRELU); print(op.kernel_instance())
L.set_option("VS_NO_WSPLIT", 1)
yd = op.forward(x, in_act=L.IN_LRELU); print(op.kernel_instance())
e = (y - yd).abs().cpu().numpy()
bad = np.argwhere(e > 1e-4)
print("bad elements", len(bad), "of", e.size)
if len(bad):
    print("batch", sorted(set(bad[:, 0]))[:8], "rows", sorted(set(bad[:, 1]))[:40], "\ncols", sorted(set(bad[:, 2]))[:80])
