#!/usr/bin/env python3
"""Census of the conv-engine launches of one step by SHAPE: (kernel instance, kind, C_in, C_out, k, dilation / stride, flags, input transform, B, T) -> count and
time -- which instances a workload needs (round 5: the tap counts / transforms the small-tile conv_ktap instances are built for).
Usage (GPU box): python tools/conv_census.py [train|infer|config5]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import ops                                 # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "train"
seen = collections.OrderedDict()
orig = ops.ConvOp.forward


def forward(self, x, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(self, x, *a, **kw)
    e1.record()
    B = kw.get("B") if kw.get("B") is not None else x.shape[0]
    T = kw.get("T") if kw.get("T") is not None else x.shape[2]
    key = (self.kernel_instance(), self.kind, self.c_in, self.c_out, self.k, self.dil, self.flags, int(kw.get("in_act", 0)), int(B), int(T),
           kw.get("res") is not None or kw.get("res_ptr") is not None, int(kw.get("out_act", 0)), int(kw.get("split_row", 0)), int(kw.get("mode", 0)))
    seen.setdefault(key, []).append((e0, e1))
    return y


if what == "train":
    from visinger_amd.models.visinger import hop256_hparams
    from visinger_amd.train import VISingerTrainer, synthetic_train_batch
    hp = hop256_hparams(p_dropout=0.1)
    torch.manual_seed(1234)
    tr = VISingerTrainer(64, 117, 131, hp).cuda().configure().train()
    batch = synthetic_train_batch(16, 512, 64, tr.hop, 64, hp["num_linear_bins"], 1234, "cuda")
    step = lambda: tr.training_step(batch)
else:
    import bench
    hidden = 512 if what == "config5" else 192
    model, hp = bench.build_model(hidden=hidden) if hidden != 192 else bench.build_model()
    model = model.cuda()
    B, T = (8, 4096) if what == "config5" else (32, 1024)
    text, pitch, dur, mel2ph, spk, noise = bench.synthetic_batch(B, T, T // 8, 64, 1234, "cuda", **({"hidden": 512} if hidden != 192 else {}))
    if what == "config5":
        from visinger_amd import _lib as L
        from visinger_amd.modules.hipconv import set_activation_storage, set_conv_math
        set_conv_math(model, L.MATH_BF16)
        set_activation_storage(model, torch.bfloat16)

    def step():
        with torch.no_grad():
            return model(text, pitch, dur, mel2ph, spk_id=spk, infer=True, noise=noise)

for _ in range(2):
    step()
torch.cuda.synchronize()
ops.ConvOp.forward = forward
step()
torch.cuda.synchronize()
ops.ConvOp.forward = orig
rows = []
for key, evs in seen.items():
    ms = sum(a.elapsed_time(b) for a, b in evs)
    rows.append((ms, len(evs), key))
tot = sum(r[0] for r in rows)
print(f"{what}: {sum(r[1] for r in rows)} conv launches, {tot:.2f} ms (event time, launch gaps included)")
print(f"{'ms':>7s} {'n':>4s} {'us':>7s}  kernel / kind cin cout k dil flags in_act B T res out_act split mode")
for ms, n, key in sorted(rows, key=lambda r: -r[0])[:70]:
    print(f"{ms:7.2f} {n:4d} {ms / n * 1e3:7.1f}  {key[0]:38s} " + " ".join(str(int(v)) for v in key[1:]))
