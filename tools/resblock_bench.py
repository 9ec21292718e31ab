#!/usr/bin/env python3
"""The whole-resblock launch (csrc/resblock_f16.hip, VS_MATH_SPLIT3) against an fp64 torch resblock (parity) and against the
one-launch-per-conv / one-launch-per-pair forms (time) at the generator's production shapes.  GPU only.
   python tools/resblock_bench.py [check|time|all|bf16]
(bf16: resblock_bf16_kernel -- plain bf16 operands on bf16-resident tensors, BASELINE config 5 -- against fp64 and against the per-pair /
per-conv bf16 launches, parity and time)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.modules.hipconv import set_conv_math  # noqa: E402
from visinger_amd.modules.visinger.decoder import ResBlock1  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"


def block(C, k, seed=0):
    torch.manual_seed(seed)
    m = ResBlock1(C, k, (1, 3, 5))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("weight_g"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) / (p.shape[1] * p.shape[2]) ** 0.5)
    return m.cuda().eval()


def ref64(m, x, acc=None, scale=1.0):
    x = x.double()
    for c1, c2 in zip(m.convs1, m.convs2):
        def w(c):
            v, g = c.weight_v.double(), c.weight_g.double()
            return g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
        xt = F.conv1d(F.leaky_relu(x, 0.1), w(c1), c1.bias.double(), padding=c1.padding[0], dilation=c1.dilation[0])
        xt = F.conv1d(F.leaky_relu(xt, 0.1), w(c2), c2.bias.double(), padding=c2.padding[0])
        x = xt + x
    if acc is not None:
        x = x + acc.double()
    return x * scale


def run(m, x, out, first, scale, pairs):
    L.set_option("VS_RESBLOCK_PAIRS", max(pairs, 0))
    L.set_option("VS_NO_RESBLOCK_FUSED", 1 if pairs < 0 else 0)
    with torch.no_grad():
        m._run_fused(x, out, first=first, scale=scale)
    return out


if what in ("check", "all"):
    for C, k, B, T in ((32, 3, 2, 3000), (64, 3, 2, 1501), (32, 7, 1, 2048), (64, 7, 2, 777), (32, 11, 2, 1000), (64, 11, 1, 4096), (64, 5, 1, 100),
                       (32, 3, 3, 7), (64, 9, 1, 232), (128, 3, 2, 1000), (128, 7, 1, 515), (128, 11, 1, 300)):
        m = block(C, k, seed=C + k)
        set_conv_math(m, L.MATH_SPLIT3)
        x = torch.randn(B, C, T, device="cuda") * 2.0
        acc = torch.randn(B, C, T, device="cuda")
        want_a = ref64(m, x, acc, 1.0 / 3)
        want_b = ref64(m, x)
        rms = want_b.pow(2).mean().sqrt().item()
        msg = []
        for pairs in (3, 1, -1):         # whole block, pair by pair, and the unfused launches
            out = acc.clone()
            run(m, x, out, False, 1.0 / 3, pairs)
            ea = (out.double() - want_a).abs().max().item() / rms
            out2 = torch.empty_like(x)
            run(m, x, out2, True, 1.0, pairs)
            eb = (out2.double() - want_b)
            msg.append(f"pairs/launch {pairs:2d}: max {max(ea, eb.abs().max().item() / rms):.2e} rms {eb.pow(2).mean().sqrt().item() / rms:.2e} [{m.convs1[0]._op().kernel_instance()}]")
        print(f"C={C} k={k} B={B} T={T}: " + " | ".join(msg), flush=True)

if what in ("time", "all"):
    B = int(os.environ.get("RB_B", 32))
    for C, T in ((128, 65536), (64, 131072), (32, 262144)):
        for k in (3, 7, 11):
            m = block(C, k)
            x = torch.randn(B, C, T, device="cuda")
            out = torch.empty_like(x)
            res = []
            for math, pairs in ((L.MATH_SPLIT6, 0), (L.MATH_SPLIT3, -1), (L.MATH_SPLIT3, 1), (L.MATH_SPLIT3, 3)) + (((L.MATH_SPLIT3, 11), (L.MATH_SPLIT3, 13)) if C == 32 else ()):
                set_conv_math(m, math)
                L.set_option("VS_RB_TILE256", 1 if pairs > 10 else 0)
                if pairs > 10:
                    pairs -= 10
                for _ in range(2):
                    run(m, x, out, True, 1.0, pairs)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    run(m, x, out, True, 1.0, pairs)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 3
                fl = 6 * 2.0 * B * C * C * k * T
                res.append(f"{'split6' if math == 6 else 'split3'}{' tile256' if L.get_option('VS_RB_TILE256') else ''} {'as today' if pairs == 0 else ('unfused' if pairs < 0 else str(pairs) + ' pair(s)/launch')}: {ms:6.2f} ms {fl / ms / 1e9:6.1f} TF")
            print(f"resblock C={C} k={k} T={T}: " + " | ".join(res), flush=True)
    L.set_option("VS_RESBLOCK_PAIRS", 0)
    L.set_option("VS_NO_RESBLOCK_FUSED", 0)


if what == "bf16":
    from visinger_amd.modules.hipconv import set_activation_storage
    B = int(os.environ.get("RB_B", 8))
    for C, T in ((128, 65536), (64, 131072), (32, 262144)):
        for k in (3, 7, 11):
            m = block(C, k)
            set_conv_math(m, L.MATH_BF16)
            x = torch.randn(B, C, T, device="cuda").bfloat16()
            want = ref64(m, x[:1, :, :4096].float())
            rms = want.pow(2).mean().sqrt().item()
            res = []
            for label, fused, pairs, norespair in (("conv by conv", 0, 0, 1), ("respair (round 2)", 0, 0, 0), ("1 pair/launch", 1, 1, 0), ("whole block", 1, 3, 0)):
                L.set_option("VS_RESBLOCK_PAIRS", pairs)
                L.set_option("VS_NO_RESBLOCK_FUSED", 0 if fused else 1)
                L.set_option("VS_NO_RESPAIR", norespair)
                out = torch.empty_like(x)
                with torch.no_grad():
                    for _ in range(2):
                        m._run_fused(x, out, first=True, scale=1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        m._run_fused(x, out, first=True, scale=1.0)
                    e1.record()
                    torch.cuda.synchronize()
                    small = torch.empty_like(x[:1, :, :4096])
                    m._run_fused(x[:1, :, :4096].contiguous(), small, first=True, scale=1.0)
                err = (small.double() - want).pow(2).mean().sqrt().item() / rms
                ms = e0.elapsed_time(e1) / 3
                res.append(f"{label}: {ms:6.2f} ms rms err {err:.1e} [{m.convs1[0]._op().kernel_instance()}]")
            print(f"bf16 resblock C={C} k={k} B={B} T={T}: " + " | ".join(res), flush=True)
    for o in ("VS_RESBLOCK_PAIRS", "VS_NO_RESBLOCK_FUSED", "VS_NO_RESPAIR"):
        L.set_option(o, 0)
