#!/usr/bin/env python3
"""conv_pipe_kernel (persistent, software-pipelined) against conv_split_kernel<1, 8, 4, 1, 3> at the generator's wide-stage shapes
(B = 32, T_mel = 1024, hop 256: 128 channels x 65536, 256 channels x 8192), interleaved A/B in one process.  GPU only."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402

B = int(os.environ.get("PB_B", 32))
REP = int(os.environ.get("PB_REP", 5))
L.set_option("VS_CONV_MATH", 3)


def one(op, x, y, res, acc, in_act, scale):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        op.forward(x, y=y, res=res, acc=acc, in_act=in_act, scale=scale)
    e0.record()
    for _ in range(REP):
        op.forward(x, y=y, res=res, acc=acc, in_act=in_act, scale=scale)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP


tot = {0: 0.0, 1: 0.0}
for C, T in ((128, 65536), (256, 8192)):
    x = torch.randn(B, C, T, device="cuda")
    y = torch.empty_like(x)
    res, acc = torch.randn_like(x), torch.randn_like(x)
    for k in (3, 7, 11):
        for d, use_res, use_acc in ((1, True, False), (3, False, False), (1, True, True)):
            op = ConvOp(L.CONV1D, C, C, k, d, (k * d - d) // 2)
            op.set_weights(torch.randn(C, C, k, device="cuda") * 0.03, None, torch.randn(C, device="cuda"))
            ms = {}
            for nopipe in (1, 0, 1, 0):
                L.set_option("VS_PIPE", 1 - nopipe)
                t = one(op, x, y, res if use_res else None, acc if use_acc else None, L.IN_LRELU, 1.0 / 3 if use_acc else 1.0)
                ms[nopipe] = min(ms.get(nopipe, 1e9), t)
                name = op.kernel_instance()
            fl = op.algorithmic_flops(B, T)
            tot[0] += ms[0]
            tot[1] += ms[1]
            print(f"C={C:3d} T={T:5d} k={k:2d} d={d} res={int(use_res)} acc={int(use_acc)}: tile kernel {ms[1]*1e3:8.1f} us ({fl/ms[1]/1e9:6.1f} TF)   pipe {ms[0]*1e3:8.1f} us "
                  f"({fl/ms[0]/1e9:6.1f} TF)   x{ms[1]/ms[0]:.3f}", flush=True)
print(f"sum: tile {tot[1]:.2f} ms, pipe {tot[0]:.2f} ms, x{tot[1]/tot[0]:.3f}")
