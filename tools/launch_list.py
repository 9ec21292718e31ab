#!/usr/bin/env python3
"""Every profiled launch of one headline step (config 2: flow inverse + generator, B=32, T_mel=1024), grouped by
(kernel instance, algorithmic flops, algorithmic bytes): count, time, TFLOP/s and GB/s per group, heaviest first.
The grouping separates conv SITES that share a kernel instance (e.g. the k=7 and k=3 resblock convs of one stage).
Usage (GPU box): python tools/launch_list.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from visinger_amd.ops import PROFILER  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model, hp = bench.build_model()
model = model.cuda()
B, T = 32, 1024
text, pitch, dur, mel2ph, spk, noise = bench.synthetic_batch(B, T, T // 8, 64, 1234, "cuda")
with torch.no_grad():
    fmask = (mel2ph > 0).float().unsqueeze(1)
    g = model.speaker_embedding(None, spk).transpose(1, 2).contiguous()
    z_p = (noise * fmask).contiguous()


def step():
    with torch.no_grad():
        z_q = model.flow(z_p, fmask, g=g, reverse=True) * fmask
        return model.decoder(z_q, g=g).squeeze(1)


for _ in range(3):
    step()
torch.cuda.synchronize()
PROFILER.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    step()
e1.record()
torch.cuda.synchronize()
PROFILER.stop()
groups = {}
for name, fl, by, a, b in PROFILER.records:
    d = groups.setdefault((name, fl, by), [0, 0.0])
    d[0] += 1
    d[1] += a.elapsed_time(b)
tot = sum(d[1] for d in groups.values()) / steps
print(f"step {e0.elapsed_time(e1) / steps:.2f} ms, profiled launches {tot:.2f} ms")
print(f"{'kernel':42s} {'n':>3s} {'GFLOP':>8s} {'MB':>8s} {'us/launch':>10s} {'ms/step':>8s} {'TF/s':>7s} {'GB/s':>7s}")
for (name, fl, by), (n, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    per = ms / n
    print(f"{name:42s} {n // steps:3d} {fl / 1e9:8.2f} {by / 1e6:8.1f} {per * 1e3:10.1f} {ms / steps:8.2f} "
          f"{fl / per / 1e9:7.1f} {by / per / 1e6:7.0f}")
