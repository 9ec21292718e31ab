#!/usr/bin/env python3
"""Is the dominant conv POWER-bound?  Runs conv_split_kernel<1, 8, 4, 1, 3> (128 channels, k = 7, B = 32, T = 65536: 8192 tiles) on streams whose
CU mask (hipExtStreamCreateWithCUMask) enables all, a half and a quarter of the chip's CUs.  The work per launch is fixed, so with a constant
per-CU rate the time doubles per halving; a smaller ratio means the full chip runs each CU slower than a partly idle chip does (the power /
clock ceiling).  Usage (GPU box): python tools/cu_mask_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(pattern):
    words = (ctypes.c_uint32 * 8)(*([pattern] * 8))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


L.set_option("VS_CONV_MATH", 3)
B = 32
rows = []
for C, T, k in ((128, 65536, 7), (128, 65536, 11), (256, 8192, 7)):
    x = torch.randn(B, C, T, device="cuda")
    y = torch.empty_like(x)
    res = torch.randn_like(x)
    op = ConvOp(L.CONV1D, C, C, k, 1, (k - 1) // 2)
    op.set_weights(torch.randn(C, C, k, device="cuda") * 0.03, None, torch.randn(C, device="cuda"))
    torch.cuda.synchronize()
    base = None
    for label, pattern, frac in (("all CUs", 0xFFFFFFFF, 1.0), ("1/2 (0x55555555)", 0x55555555, 0.5), ("1/2 (0x0F0F0F0F)", 0x0F0F0F0F, 0.5),
                                 ("1/4 (0x11111111)", 0x11111111, 0.25), ("1/8 (0x01010101)", 0x01010101, 0.125)):
        st = masked_stream(pattern)
        with torch.cuda.stream(st):
            for _ in range(2):
                op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
            e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 3 * 1e3
        base = base or us
        print(f"C={C} k={k:2d}  {label:18s} {us:9.1f} us   x{us / base:5.2f} of the full chip's time   per-CU rate x{base / (us * frac):5.2f}", flush=True)
