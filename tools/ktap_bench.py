"""conv_ktap_kernel against conv_split_kernel<1, 8, 4, 1, 3> at the production shapes of the headline batch (B=32), interleaved A/B in one process
(VS_NO_KTAP through vs_set_option).  Usage: python tools/ktap_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L                      # noqa: E402
from visinger_amd.ops import ConvOp                     # noqa: E402

SHAPES = [  # C, k, d, T, res
    (128, 3, 1, 65536, True), (128, 7, 1, 65536, True), (128, 7, 5, 65536, False), (128, 11, 1, 65536, True), (128, 11, 5, 65536, False),
    (256, 3, 1, 8192, True), (256, 7, 3, 8192, True), (256, 11, 5, 8192, False),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    L.set_option("VS_CONV_MATH", 3)
    B = 32
    for C, k, d, T, use_res in SHAPES:
        op = ConvOp(L.CONV1D, C, C, k, d, (k * d - d) // 2)
        op.set_weights(torch.randn(C, C, k, device="cuda") * (C * k) ** -0.5, None, torch.zeros(C, device="cuda"))
        x = torch.randn(B, C, T, device="cuda")
        res = torch.randn(B, C, T, device="cuda") if use_res else None
        y = torch.empty_like(x)
        t = {}
        for rnd in range(3):
            for no in (1, 0):
                L.set_option("VS_NO_KTAP", no)
                op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
                e1.record()
                torch.cuda.synchronize()
                t.setdefault(op.kernel_instance(), []).append(e0.elapsed_time(e1) / reps * 1e3)
        fl = 2.0 * B * C * C * k * T
        line = f"C={C} k={k} d={d} T={T} res={int(use_res)}: "
        for name, v in t.items():
            us = min(v)
            line += f"{name} {us:8.1f} us {fl / us / 1e6:6.1f} TFLOP/s   "
        print(line, flush=True)


if __name__ == "__main__":
    main()
