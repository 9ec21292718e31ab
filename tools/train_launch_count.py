#!/usr/bin/env python3
"""Launch census of one config-3 training step (torch profiler): kernels by launch count and the aten / autograd-function ops that
issue them -- the step is host-launch-bound (tools/train_step_bench.py), so the count is what to cut.  Usage (GPU box):
python tools/train_launch_count.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd.models.visinger import hop256_hparams  # noqa: E402
from visinger_amd.train import VISingerTrainer, synthetic_train_batch  # noqa: E402

hp = hop256_hparams(p_dropout=0.1)
torch.manual_seed(1234)
tr = VISingerTrainer(64, 117, 131, hp).cuda().configure().train()
batch = synthetic_train_batch(16, 512, 64, tr.hop, 64, hp["num_linear_bins"], 1234, "cuda")
for _ in range(3):
    tr.training_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tr.training_step(batch)
    torch.cuda.synchronize()
ev = prof.key_averages()
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
print(f"kernel launches: {sum(e.count for e in kern)}, device time {sum(e.device_time_total for e in kern) / 1e3:.1f} ms")
print("--- kernels by count")
for e in sorted(kern, key=lambda e: -e.count)[:30]:
    print(f"{e.count:6d} {e.device_time_total / 1e3:8.2f} ms  {e.key[:110]}")
print("--- kernels by device time")
for e in sorted(kern, key=lambda e: -e.device_time_total)[:40]:
    print(f"{e.count:6d} {e.device_time_total / 1e3:8.2f} ms  {e.key[:110]}")
print("--- host ops by count")
for e in sorted(cpu, key=lambda e: -e.count)[:25]:
    print(f"{e.count:6d} {e.cpu_time_total / 1e3:8.2f} ms (self {e.self_cpu_time_total / 1e3:7.2f})  {e.key[:90]}")
