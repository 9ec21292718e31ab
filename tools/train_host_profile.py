#!/usr/bin/env python3
"""Where the HOST time of a config-3 training step goes: cProfile over a few steps (the step is host-launch-bound: tools/train_step_bench.py), top
functions by own time, plus the step's wall time against the device time of its kernels.  Usage (GPU box): python tools/train_host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd.models.visinger import hop256_hparams  # noqa: E402
from visinger_amd.train import VISingerTrainer, synthetic_train_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hp = hop256_hparams(p_dropout=0.1)
torch.manual_seed(1234)
tr = VISingerTrainer(64, 117, 131, hp).cuda().configure().train()
batch = synthetic_train_batch(16, 512, 64, tr.hop, 64, hp["num_linear_bins"], 1234, "cuda")
for _ in range(4):
    tr.training_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.training_step(batch)
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step (un-profiled)")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    tr.training_step(batch)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
