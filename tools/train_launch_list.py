#!/usr/bin/env python3
"""Conv-engine launches of one config-3 training step grouped by (kernel instance, algorithmic flops, bytes): count, time, TFLOP/s,
GB/s -- which conv sites the step's engine time goes to.  Usage (GPU box): python tools/train_launch_list.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd.models.visinger import hop256_hparams  # noqa: E402
from visinger_amd.ops import PROFILER  # noqa: E402
from visinger_amd.train import VISingerTrainer, synthetic_train_batch  # noqa: E402

hp = hop256_hparams(p_dropout=0.1)
torch.manual_seed(1234)
tr = VISingerTrainer(64, 117, 131, hp).cuda().configure().train()
batch = synthetic_train_batch(16, 512, 64, tr.hop, 64, hp["num_linear_bins"], 1234, "cuda")
for _ in range(3):
    tr.training_step(batch)
torch.cuda.synchronize()
PROFILER.start()
tr.training_step(batch)
torch.cuda.synchronize()
PROFILER.stop()
groups = {}
for name, fl, by, a, b in PROFILER.records:
    d = groups.setdefault((name, fl, by), [0, 0.0])
    d[0] += 1
    d[1] += a.elapsed_time(b)
print(f"profiled engine launches: {len(PROFILER.records)}, {sum(d[1] for d in groups.values()):.2f} ms")
print(f"{'kernel':42s} {'n':>4s} {'GFLOP':>8s} {'MB':>8s} {'us/launch':>10s} {'ms/step':>8s} {'TF/s':>7s} {'GB/s':>7s}")
for (name, fl, by), (n, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:60]:
    per = ms / n
    print(f"{name:42s} {n:4d} {fl / 1e9:8.3f} {by / 1e6:8.1f} {per * 1e3:10.1f} {ms:8.2f} {fl / per / 1e9:7.1f} {by / per / 1e6:7.0f}")
