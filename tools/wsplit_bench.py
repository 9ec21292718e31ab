#!/usr/bin/env python3
"""A/B of the F(2,3)-on-split kernel (csrc/conv_wsplit.hip) against the direct split-bf16 x6 kernel at the production shapes of the
128- / 256-channel generator stages and the FFN (B=32, T_mel=1024, hop 256), interleaved rounds in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp

B = int(os.environ.get("CB_B", 32))
L.set_option("VS_WSPLIT_FORCE", 1)


def run(op, x, y, res, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_w = tot_d = 0.0
for cin, cout, T in ((256, 256, 8192), (128, 128, 65536), (64, 64, 131072), (192, 768, 1024)):
    for k in ((3, 7, 11) if cin == cout else (9,)):
        for d in ((1, 3, 5) if cin == cout else (1,)):
            op = ConvOp(L.CONV1D, cin, cout, k, d, d * (k - 1) // 2)
            op.set_weights(torch.randn(cout, cin, k, device="cuda") * 0.05, None, torch.randn(cout, device="cuda"))
            x = torch.randn(B, cin, T, device="cuda")
            y = torch.empty(B, cout, T, device="cuda")
            res = torch.randn_like(y)
            tw, td = [], []
            for rnd in range(3):
                L.set_option("VS_NO_WSPLIT", 0)
                run(op, x, y, res, 1)
                tw.append(run(op, x, y, res, 4))
                kw = op.kernel_instance()
                L.set_option("VS_NO_WSPLIT", 1)
                run(op, x, y, res, 1)
                td.append(run(op, x, y, res, 4))
                kd = op.kernel_instance()
            L.set_option("VS_NO_WSPLIT", 0)
            fl = op.algorithmic_flops(B, T)
            w, dd = min(tw), min(td)
            mult = (4 if d == 1 else 1) if cin == cout else 1
            tot_w += w * mult; tot_d += dd * mult
            print(f"{cin}->{cout} k={k} d={d} T={T}: {kw} {w:8.1f} us {fl/w/1e6:6.1f} TF | {kd} {dd:8.1f} us {fl/dd/1e6:6.1f} TF | x{dd/w:.2f}", flush=True)
print(f"weighted total (6 convs per k and stage): F(2,3)-split {tot_w/1e3:.2f} ms, direct {tot_d/1e3:.2f} ms")
