#!/usr/bin/env python3
"""Per-workgroup phase timeline of one whole-resblock / pair launch (csrc/resblock_f16.hip, vs_debug_set_stamp_buffer): x load, and per conv the input
transform + exponent barrier, the tile split / write, the MFMA loop, the scale-out; epilogue.  Env: C (64), K (7), T (131072), B (32), PAIRS (as the
production table chooses when unset: 1 = pair by pair, 3 = whole block), PAIR (which pair of the block to time when PAIRS=1: 0..2)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L                                   # noqa: E402
from visinger_amd.modules.hipconv import set_conv_math               # noqa: E402
from visinger_amd.modules.visinger.decoder import ResBlock1          # noqa: E402

C, k, T, B = (int(os.environ.get(n, d)) for n, d in (("C", 64), ("K", 7), ("T", 131072), ("B", 32)))
pairs = int(os.environ.get("PAIRS", 1))
torch.manual_seed(0)
m = ResBlock1(C, k, (1, 3, 5)).cuda().eval()
set_conv_math(m, L.MATH_SPLIT3)
x = torch.randn(B, C, T, device="cuda")
out = torch.empty_like(x)
L.set_option("VS_RESBLOCK_PAIRS", pairs)
with torch.no_grad():
    for _ in range(2):
        m._run_fused(x, out, first=True, scale=1.0)
    torch.cuda.synchronize()
    lib = L.lib()
    lib.vs_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    nlaunch = 3 // pairs
    for li in range(nlaunch):
        buf = torch.zeros(65536 * 64, dtype=torch.int64, device="cuda")
        # stamp only launch `li` of the block: the hook is global, so run the block with the buffer set and keep the LAST writer per slot -- simpler: time pair li alone
        convs = [c for pr in zip(m.convs1, m.convs2) for c in pr][2 * pairs * li: 2 * pairs * (li + 1)]
        from visinger_amd.ops import resblock_forward                # noqa: E402
        hs = [c._op() for c in convs]
        resblock_forward(hs, x, out)                                  # warm
        torch.cuda.synchronize()
        lib.vs_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
        resblock_forward(hs, x, out)
        torch.cuda.synchronize()
        lib.vs_debug_set_stamp_buffer(None)
        full = buf.cpu().numpy().reshape(-1, 64)
        full = full[full[:, 0] != 0]
        nconv = len(convs)
        end = 2 + 4 * nconv
        t0 = full[:, 0].min()
        print(f"launch {li}: {hs[0].last_kernel()}  dil {[c.dilation[0] for c in convs]}  workgroups {len(full)}  span {(full[:, end].max() - t0) / 100.0:.1f} us")
        names = ["x load"] + sum([[f"c{c} transform+bar", f"c{c} split/write+bar", f"c{c} MFMA loop", f"c{c} scale-out"] for c in range(nconv)], []) + ["epilogue"]
        prev = full[:, 0]
        tot = (full[:, end] - full[:, 0]) / 100.0
        for i, nm in enumerate(names):
            slot = i + 1 if i + 1 < end else end
            d = (full[:, slot] - prev) / 100.0
            cyc = (full[:, 32 + slot] - full[:, 32 + (slot - 1 if i > 0 else 0)]).astype(np.float64)
            print(f"   {nm:22s} mean {d.mean():7.2f} p10 {np.percentile(d, 10):7.2f} p90 {np.percentile(d, 90):7.2f} us   {cyc.mean():9.0f} cycles")
            prev = full[:, slot]
        print(f"   {'total':22s} mean {tot.mean():7.2f} p10 {np.percentile(tot, 10):7.2f} p90 {np.percentile(tot, 90):7.2f} us; shader clock {((full[:, 32 + end] - full[:, 32]) / (tot * 1e3)).mean():.3f} GHz")
