#!/usr/bin/env python3
"""Three launches each of the dominant conv instances (fp32 tensors / split-f16 x3; bf16-resident tensors / plain bf16) for an instruction-mix counter pass:
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES --kernel-trace --output-format csv -d out -- python3 tools/inst_mix.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402

for math, dt, B, C, T, k in ((3, torch.float32, 32, 128, 65536, 7), (1, torch.bfloat16, 8, 128, 262144, 7)):
    L.set_option("VS_CONV_MATH", math)
    x = torch.randn(B, C, T, device="cuda").to(dt)
    y = torch.empty_like(x)
    res = torch.randn(B, C, T, device="cuda").to(dt)
    op = ConvOp(L.CONV1D, C, C, k, 1, (k - 1) // 2)
    op.set_weights(torch.randn(C, C, k, device="cuda") * 0.03, None, torch.randn(C, device="cuda"))
    for _ in range(3):
        op.forward(x, y=y, res=res, in_act=L.IN_LRELU)
    torch.cuda.synchronize()
    print(op.kernel_instance(), "tiles", B * T // 256, "steps per tile", (C // 16) * k)
