#!/usr/bin/env python3
"""Randomised parity sweep of the conv engine (split-bf16, fp32 direct + F(2,3), small-C_out, transposed paths) against the
fp64 oracle; the arithmetic (vs_conv_math) of each case is drawn at random unless VS_CONV_MATH pins it.
    python tools/conv_fuzz.py [n_cases] [seed]
GPU only; prints the worst scaled error per kernel instance and fails on the first case above tolerance."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import visinger_oracle as orc  # noqa: E402  (checker)
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.ops import ConvOp  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    r = np.random.default_rng(seed)
    worst = {}
    for case in range(n):
        transposed = r.random() < 0.2
        math = int(L.get_option("VS_CONV_MATH")) if L.get_option("VS_CONV_MATH") >= 0 else int(r.choice([L.MATH_F32, L.MATH_SPLIT6]))
        B = int(r.integers(1, 4))
        if transposed:
            u = int(r.choice([2, 3, 4, 5, 8]))
            k = int(u * r.integers(1, 3) + r.choice([0, 1]) * (u % 2))
            k = max(k, u)
            if (k - u) % 2:
                k += 1
            Cin, Cout, T = int(r.choice([8, 16, 32, 48, 64, 128])), int(r.choice([4, 8, 16, 32, 64])), int(r.integers(1, 200))
            pad = (k - u) // 2
            x = r.standard_normal((B, Cin, T)).astype(np.float32)
            w = (r.standard_normal((Cin, Cout, k)) / np.sqrt(Cin * k / u)).astype(np.float32)
            bias = r.standard_normal(Cout).astype(np.float32)
            ref = orc.conv_transpose1d(orc.leaky_relu(x.astype(np.float64)), w, bias, stride=u, padding=pad)
            op = ConvOp(L.CONV_TRANSPOSE1D, Cin, Cout, k, u, pad).set_math(math)
            op.set_weights(dev(w), None, dev(bias))
            y = op.forward(dev(x), in_act=L.IN_LRELU)
            desc = f"tconv B{B} {Cin}->{Cout} k{k} u{u} T{T}"
        else:
            k = int(r.choice([1, 3, 5, 7, 9, 11]))
            d = int(r.choice([1, 1, 3, 5])) if k > 1 else 1
            if (k - 1) * d > 60:
                d = 1
            Cin = int(r.choice([1, 7, 16, 32, 33, 64, 96, 128, 192, 256]))
            Cout = int(r.choice([1, 2, 16, 32, 64, 96, 128, 192, 256, 70]))
            T = int(r.choice([1, 2, 5, 31, 64, 100, 127, 128, 129, 255, 256, 257, 500, 512, 1000, 1024, 2047]))
            pad = d * (k - 1) // 2
            forced = r.random() < 0.5
            L.set_option("VS_WINO_FORCE", int(forced))
            x = r.standard_normal((B, Cin, T)).astype(np.float32)
            w = (r.standard_normal((Cout, Cin, k)) / np.sqrt(Cin * k)).astype(np.float32)
            bias = r.standard_normal(Cout).astype(np.float32) if r.random() < 0.8 else None
            mask = (np.arange(T)[None] < r.integers(1, T + 1, (B, 1))).astype(np.float32)
            use_res, use_acc = (r.random() < 0.5), (r.random() < 0.3)
            if Cout <= 4:
                use_res = use_acc = False
            in_act = int(r.choice([L.IN_NONE, L.IN_LRELU, L.IN_MASK, L.IN_LRELU_MASK]))
            out_act = int(r.choice([L.OUT_NONE, L.OUT_NONE, L.OUT_TANH, L.OUT_RELU]))
            out_mask = bool(r.random() < 0.3)
            scale = float(r.choice([1.0, 1.0, 1.0 / 3.0]))
            res = r.standard_normal((B, Cout, T)).astype(np.float32)
            acc = r.standard_normal((B, Cout, T)).astype(np.float32)
            xin = x.astype(np.float64)
            if in_act in (L.IN_LRELU, L.IN_LRELU_MASK):
                xin = orc.leaky_relu(xin)
            if in_act in (L.IN_MASK, L.IN_LRELU_MASK):
                xin = xin * mask[:, None]
            ref = orc.conv1d(xin, w, bias, dilation=d, padding=pad)
            if use_res:
                ref = ref + res
            if use_acc:
                ref = ref + acc
            ref = ref * scale
            ref = np.tanh(ref) if out_act == L.OUT_TANH else (np.maximum(ref, 0) if out_act == L.OUT_RELU else ref)
            if out_mask:
                ref = ref * mask[:, None]
            op = ConvOp(L.CONV1D, Cin, Cout, k, d, pad).set_math(math)
            op.set_weights(dev(w), None, None if bias is None else dev(bias))
            y = op.forward(dev(x), in_act=in_act, mask=dev(mask), res=dev(res) if use_res else None, acc=dev(acc) if use_acc else None,
                           scale=scale, out_act=out_act, out_mask=out_mask)
            desc = (f"conv B{B} {Cin}->{Cout} k{k} d{d} T{T} in{in_act} out{out_act} res{int(use_res)} acc{int(use_acc)} "
                    f"scale{scale:.2f} omask{int(out_mask)} forced{int(forced)}")
        torch.cuda.synchronize()
        got = y.cpu().double().numpy()
        assert got.shape == ref.shape, (desc, got.shape, ref.shape)
        err = float((np.abs(got - ref) / (1.0 + np.abs(ref))).max()) if got.size else 0.0
        inst = op.kernel_instance()
        if err > worst.get(inst, (0.0, ""))[0]:
            worst[inst] = (err, desc)
        if not np.isfinite(got).all() or err > (3e-5 if math != L.MATH_BF16 else 3e-2):
            print("FAIL", desc, inst, "err", err)
            sys.exit(1)
    for k_, (e, dsc) in sorted(worst.items()):
        print(f"{k_:32s} worst scaled err {e:.2e}  ({dsc})")
    print(f"OK {n} cases")


if __name__ == "__main__":
    main()
