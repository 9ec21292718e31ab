#!/usr/bin/env python3
"""Where a BASELINE config-3 training step spends its time: GPU time (HIP events on the current stream) and host time per phase of
VISingerTrainer.training_step -- generator pass forward (model / mel losses / discriminators), its backward, optimizer; discriminator pass
forward, backward, optimizer.  GPU only.   python tools/train_phases.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.models.visinger import hop256_hparams  # noqa: E402
from visinger_amd.train import VISingerTrainer, synthetic_train_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
hp = hop256_hparams(p_dropout=0.1)
torch.manual_seed(1234)
tr = VISingerTrainer(64, 117, 131, hp).to(dev).configure().train()
batch = synthetic_train_batch(16, 512, 64, tr.hop, 64, hp["num_linear_bins"], 1234, dev)
for _ in range(3):
    tr.training_step(batch)
torch.cuda.synchronize()

marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e, time.perf_counter()))


# the pieces of generator_pass, re-stated with marks between them (same calls as train.py)
orig_model_forward = tr.model.forward
orig_disc_forward = tr.mel_disc.forward


def model_forward(*a, **k):
    mark("g: before model forward")
    out = orig_model_forward(*a, **k)
    mark("g: model forward (prior + posterior + flow + segment decode)")
    return out


def disc_forward(*a, **k):
    mark("before discriminators")
    out = orig_disc_forward(*a, **k)
    mark("discriminators forward")
    return out


tr.model.forward = model_forward
tr.mel_disc.forward = disc_forward
totals = {}
host = {}
wall0 = time.perf_counter()
for s in range(steps):
    marks.clear()
    mark("step start")
    for opt_idx, opt in enumerate((tr.opt_gen, tr.opt_disc)):
        tag = "G" if opt_idx == 0 else "D"
        own, other = (tr.model, tr.mel_disc) if opt_idx == 0 else (tr.mel_disc, tr.model)
        for p in other.parameters():
            p.requires_grad_(False)
        for p in own.parameters():
            p.requires_grad_(True)
        mark(f"{tag}: pass start")
        loss, parts = tr(batch, opt_idx)
        mark(f"{tag}: losses (mel spectrograms, kl, ctc, gan)")
        loss.backward()
        mark(f"{tag}: backward")
        torch.nn.utils.clip_grad_norm_(tr.parameters(), tr.hp["clip_grad_norm"])
        opt.step()
        opt.zero_grad(set_to_none=True)
        tr.on_after_optimization()
        mark(f"{tag}: clip + AdamW")
        _ = {k: float(v.detach()) for k, v in parts.items()}
        mark(f"{tag}: loss values to host")
    for p in tr.parameters():
        p.requires_grad_(True)
    tr.global_step += 1
    torch.cuda.synchronize()
    for (n0, e0, h0), (n1, e1, h1) in zip(marks, marks[1:]):
        key = n1
        totals[key] = totals.get(key, 0.0) + e0.elapsed_time(e1)
        host[key] = host.get(key, 0.0) + (h1 - h0) * 1e3
wall = (time.perf_counter() - wall0) / steps * 1e3
print(f"{steps} steps, {wall:.1f} ms per step (wall)")
order = []
for n, _, _ in marks[1:]:
    if n not in order:
        order.append(n)
tot = 0.0
for n in order:
    print(f"  {n:70s} GPU {totals[n] / steps:7.2f} ms   host {host[n] / steps:7.2f} ms")
    tot += totals[n] / steps
print(f"  {'sum':70s} GPU {tot:7.2f} ms")
