#!/usr/bin/env python3
"""BASELINE config 3: full GAN training step (generator pass + discriminator pass, both optimizers) on one MI355X,
synthetic B=16, T_mel=512, segment 32 frames, hop 256, reference-size networks.  Prints ms/step and the split."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd.models.visinger import hop256_hparams  # noqa: E402
from visinger_amd.train import VISingerTrainer, synthetic_train_batch  # noqa: E402

B, T = int(os.environ.get("TB_B", 16)), int(os.environ.get("TB_T", 512))
hp = hop256_hparams(p_dropout=float(os.environ.get("TB_DROPOUT", 0.1)))      # config/models/visinger.yaml:9
torch.manual_seed(1234)
tr = VISingerTrainer(64, 117, 131, hp).cuda().configure().train()
batch = synthetic_train_batch(B, T, T // 8, tr.hop, 64, hp["num_linear_bins"], 1234, "cuda")


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_first = timed(lambda: tr.training_step(batch), 1)
timed(lambda: tr.training_step(batch), 2)
ms = timed(lambda: tr.training_step(batch), 5)
print(f"first step {t_first:.0f} ms (includes lazy kernel selection); steady state {ms:.1f} ms/step  "
      f"= {B * tr.segment_size * tr.hop / ms * 1e3:.3g} generated samples/s, {B * T / ms * 1e3:.3g} frames/s", flush=True)


def gen_fwd():
    for p in tr.mel_disc.parameters():
        p.requires_grad_(False)
    loss, _ = tr.generator_pass(batch)
    return loss


fwd = timed(gen_fwd, 3)
loss = gen_fwd()
bwd = timed(lambda: torch.autograd.grad(loss, [p for p in tr.model.parameters() if p.requires_grad], retain_graph=True,
                                         allow_unused=True), 3)
for p in tr.parameters():
    p.requires_grad_(True)
print(f"generator pass: forward {fwd:.1f} ms, backward {bwd:.1f} ms", flush=True)
if os.environ.get("TB_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        tr.training_step(batch)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
