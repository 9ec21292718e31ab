#!/usr/bin/env python3
"""Which Python lines of the training step launch its aten kernels: one config-3 step under torch.profiler (with stacks); for each
(kernel name, innermost repo source line) the launches and device time per step.  Usage (GPU box): python tools/train_op_census.py [min_launches]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visinger_amd.models.visinger import hop256_hparams  # noqa: E402
from visinger_amd.train import VISingerTrainer, synthetic_train_batch  # noqa: E402

min_launches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hp = hop256_hparams(p_dropout=0.1)
torch.manual_seed(1234)
tr = VISingerTrainer(64, 117, 131, hp).cuda().configure().train()
batch = synthetic_train_batch(16, 512, 64, tr.hop, 64, hp["num_linear_bins"], 1234, "cuda")
for _ in range(3):
    tr.training_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.training_step(batch)
    torch.cuda.synchronize()


def site(ev):
    """innermost frame of the repo in the stack of the CPU op that launched `ev` (outermost aten op's stack)"""
    for fr in ev.stack or ():
        if "visinger_amd" in fr and "site-packages" not in fr:
            return fr.replace(ROOT + "/", "").strip()
    return (ev.stack[0].strip() if ev.stack else "?")


by_site = collections.defaultdict(lambda: [0, 0.0])
by_op = collections.defaultdict(lambda: [0, 0.0])
n_kern = 0
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.kernels:
        # only ops that launch directly (children launch for their parents too: count at the leaf)
        leaf = not any(ch.kernels for ch in ev.cpu_children)
        if not leaf:
            continue
        dev_us = sum(k.duration for k in ev.kernels)
        n = len(ev.kernels)
        n_kern += n
        # the stack of the outermost parent carries the Python frames
        top = ev
        while top.cpu_parent is not None:
            top = top.cpu_parent
        s = site(top) if top.stack else site(ev)
        if s == "?":
            s = top.name
        d = by_site[(ev.name, s)]
        d[0] += n
        d[1] += dev_us
        d = by_op[ev.name]
        d[0] += n
        d[1] += dev_us
print(f"kernels attributed: {n_kern}")
print("---- by op")
for name, (n, us) in sorted(by_op.items(), key=lambda kv: -kv[1][0])[:50]:
    print(f"{n:6d} launches {us / 1e3:8.2f} ms  {name}")
print("---- by (op, site)")
for (name, s), (n, us) in sorted(by_site.items(), key=lambda kv: -kv[1][0]):
    if n < min_launches:
        break
    print(f"{n:6d} launches {us / 1e3:8.2f} ms  {name:40s} {s}")
