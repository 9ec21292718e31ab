#!/usr/bin/env python3
"""Micro-benchmark of the conv engine at the generator's production shapes (B=32, T_mel=1024, hop 256).
Prints achieved TFLOP/s per shape (algorithmic FLOPs / HIP-event time).  GPU only."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402

B = int(os.environ.get("CB_B", 32))
REP = int(os.environ.get("CB_REP", 5))


def bench(name, kind, cin, cout, k, d, T, **kw):
    pad = (k * d - d) // 2 if kind != L.CONV_TRANSPOSE1D else (k - d) // 2
    op = ConvOp(kind, cin, cout, k, d, pad)
    w = torch.randn((cout, cin, k) if kind != L.CONV_TRANSPOSE1D else (cin, cout, k), device="cuda") * 0.05
    op.set_weights(w, None, torch.randn(cout, device="cuda"))
    x = torch.randn(B, cin, T, device="cuda")
    Tout = op.out_len(T)
    y = torch.empty(B, op.rows_out, Tout, device="cuda")
    res = torch.randn_like(y) if kw.pop("res", False) else None
    for _ in range(2):
        op.forward(x, y=y, res=res, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        op.forward(x, y=y, res=res, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REP
    fl = op.algorithmic_flops(B, T)
    print(f"{name:34s} {op.kernel_instance():28s} {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  ({fl/1e9:7.1f} GFLOP)", flush=True)
    return ms


if __name__ == "__main__":
    tot = 0.0
    for C, T in ((256, 8192), (128, 65536), (64, 131072), (32, 262144)):
        for k in (3, 7, 11):
            for d in (1, 3, 5):
                ms = bench(f"resblock C={C} k={k} d={d}", L.CONV1D, C, C, k, d, T, in_act=L.IN_LRELU, res=True)
                tot += ms * (4 if d == 1 else 1)   # 6 convs per (stage, k): convs1 d=1,3,5 + convs2 d=1 x3
    bench("ups0 512->256 k16 u8", L.CONV_TRANSPOSE1D, 512, 256, 16, 8, 1024, in_act=L.IN_LRELU)
    bench("ups1 256->128 k16 u8", L.CONV_TRANSPOSE1D, 256, 128, 16, 8, 8192, in_act=L.IN_LRELU)
    bench("ups2 128->64 k4 u2", L.CONV_TRANSPOSE1D, 128, 64, 4, 2, 65536, in_act=L.IN_LRELU)
    bench("ups3 64->32 k4 u2", L.CONV_TRANSPOSE1D, 64, 32, 4, 2, 131072, in_act=L.IN_LRELU)
    bench("conv_pre 192->512 k7", L.CONV1D, 192, 512, 7, 1, 1024)
    bench("conv_post 32->1 k7", L.CONV1D, 32, 1, 7, 1, 262144, in_act=L.IN_LRELU, out_act=L.OUT_TANH)
    bench("1x1 192->192 T=1024", L.CONV1D, 192, 192, 1, 1, 1024)
    bench("1x1 192->384 T=1024", L.CONV1D, 192, 384, 1, 1, 1024)
    bench("ffn1 192->768 k9", L.CONV1D, 192, 768, 9, 1, 1024, out_act=L.OUT_RELU)
    bench("ffn2 768->192 k9", L.CONV1D, 768, 192, 9, 1, 1024)
    bench("wavenet in 192->384 k5 (gate)", L.CONV1D_PAIRED, 192, 384, 5, 1, 1024, pair_mode=L.PAIR_GATE)
    print(f"approx resblock total {tot:.1f} ms")
