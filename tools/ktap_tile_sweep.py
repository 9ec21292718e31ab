#!/usr/bin/env python3
"""Which tile shape should a short conv launch take?  Times the conv_ktap instances of each tile (VS_CONV_CFG: 0 = 128 x 256, 3 = 64 x 256, 2 = 32 x 256, 6 = 32 x 128) on the
shapes the training step and the T_mel-sized transformer convs launch (tools/conv_census.py), next to what vs_conv_forward chooses by itself (-1).
Usage: python tools/ktap_tile_sweep.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L                      # noqa: E402
from visinger_amd.ops import ConvOp                     # noqa: E402

SHAPES = [  # C_in, C_out, k, flags, in_act (0 none / 2 mask), B, T
    (384, 192, 5, 4, 0, 16, 512), (192, 384, 5, 0, 0, 16, 512), (768, 192, 9, 4, 0, 16, 512), (192, 768, 9, 0, 0, 16, 512),
    (192, 384, 1, 0, 0, 16, 512), (384, 192, 1, 4, 0, 16, 512), (768, 192, 1, 0, 0, 16, 512), (192, 192, 1, 0, 0, 16, 512), (576, 192, 1, 4, 0, 16, 512),
    (1024, 1024, 5, 0, 0, 1, 2468), (1024, 1024, 5, 0, 0, 1, 4932), (1536, 1024, 2, 0, 0, 1, 1937), (1536, 1024, 2, 0, 0, 1, 3873),
    (256, 256, 11, 0, 0, 16, 256), (128, 128, 11, 0, 0, 16, 2048), (128, 128, 7, 4, 0, 16, 2048), (768, 192, 9, 4, 0, 16, 64),
    (768, 192, 1, 0, 2, 32, 1024), (192, 576, 1, 0, 0, 32, 1024), (192, 192, 1, 0, 0, 32, 1024), (192, 768, 9, 0, 2, 32, 128), (192, 384, 1, 0, 0, 32, 1024),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    L.set_option("VS_CONV_MATH", 3)
    for cin, cout, k, flags, ia, B, T in SHAPES:
        adj = flags == 4
        op = ConvOp(L.CONV1D, cin, cout, k, 1, k // 2, flags)
        w = torch.randn(*((cin, cout, k) if adj else (cout, cin, k)), device="cuda") * (cin * k) ** -0.5
        op.set_weights(w, None, None if adj else torch.zeros(cout, device="cuda"))
        x = torch.randn(B, cin, T, device="cuda")
        mask = torch.ones(B, T, device="cuda") if ia else None
        line = f"{cin:5d}->{cout:<5d} k{k:<2d} f{flags} a{ia} B{B:<2d} T{T:<5d}:"
        best = None
        for cfg in (-1, 0, 3, 2, 6):
            L.set_option("VS_CONV_CFG", cfg)
            try:
                op.forward(x, in_act=ia, mask=mask)
            except L.VisingerHipError:
                line += f"  cfg{cfg}: -"
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                e0.record()
                for _ in range(reps):
                    op.forward(x, in_act=ia, mask=mask)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps * 1e3)
            us = min(ts)
            name = op.kernel_instance()
            short = name.replace("conv_ktap_kernel", "kt").replace("conv_split_kernel", "cs")
            line += f"  cfg{cfg}: {us:6.1f} us [{short}]"
            if cfg >= 0 and (best is None or us < best[0]):
                best = (us, cfg)
        print(line + f"   best cfg {best[1]}", flush=True)
    L.set_option("VS_CONV_CFG", -1)


if __name__ == "__main__":
    main()
