#!/usr/bin/env python3
"""Training-mode attention core (csrc/attention_train.hip: AttnCoreFn) against the PyTorch [T, T] version of the same function
(autograd.attention with VS_NO_TRAIN_ATTN), forward and forward + backward, at the config-3 shape B=16, 2 heads x 96, T=512.
Usage (GPU box): python tools/attn_train_bench.py [B T p_drop]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import autograd as A  # noqa: E402
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.modules.rel_transformer import MultiHeadAttention  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
pd = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
torch.manual_seed(0)
m = MultiHeadAttention(192, 192, 2, window_size=4, p_dropout=pd).cuda().train()
x = torch.randn(B, 192, T).cuda().requires_grad_(True)
fm = torch.ones(B, T).cuda()
q, k, v = (torch.randn(B, 192, T).cuda().requires_grad_(True) for _ in range(3))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def core_fwd():
    return A.AttnCoreFn.apply(q, k, v, m.emb_rel_k, m.emb_rel_v, fm, 2, 4, pd)


def core_fb():
    torch.autograd.grad(core_fwd().sum(), [q, k, v, m.emb_rel_k, m.emb_rel_v])


def layer_fb():
    y = A.attention(m, x, fm)
    torch.autograd.grad(y.sum(), [x] + [p for p in m.parameters()], allow_unused=True)


print(f"B={B} T={T} p_drop={pd}")
print(f"AttnCoreFn forward {timed(core_fwd):.3f} ms, forward + backward {timed(core_fb):.3f} ms")
print(f"whole attention layer (q/k/v/o convs + core), forward + backward: HIP core {timed(layer_fb):.3f} ms", end="")
L.set_option("VS_NO_TRAIN_ATTN", 1)
print(f", PyTorch [T,T] core {timed(layer_fb):.3f} ms")
