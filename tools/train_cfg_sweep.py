#!/usr/bin/env python3
"""Which tile shape should each conv-engine launch of the config-3 training step take?  Runs the step under several forced
tile-shape settings (the engine's A/B switches, set through vs_set_option), lines the profiled launches up by index (the call
sequence of a step is fixed) and prints, per launch group, the time under each setting and what a per-launch best choice would
save.  Usage (GPU box): python tools/train_cfg_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L  # noqa: E402
from visinger_amd.models.visinger import hop256_hparams  # noqa: E402
from visinger_amd.ops import PROFILER  # noqa: E402
from visinger_amd.train import VISingerTrainer, synthetic_train_batch  # noqa: E402

SETTINGS = [
    ("default", {}),
    ("t6=0 (32x256)", {"VS_SMALL_GRID_T6": 0}),
    ("t6=inf (32x128)", {"VS_SMALL_GRID_T6": 1 << 30}),
    ("no_small_grid", {"VS_NO_SMALL_GRID": 1}),
    ("cfg=3 (64x256)", {"VS_CONV_CFG": 3}),
    ("cfg=0 (128x256)", {"VS_CONV_CFG": 0}),
]

hp = hop256_hparams(p_dropout=0.1)
torch.manual_seed(1234)
tr = VISingerTrainer(64, 117, 131, hp).cuda().configure().train()
batch = synthetic_train_batch(16, 512, 64, tr.hop, 64, hp["num_linear_bins"], 1234, "cuda")
for _ in range(3):
    tr.training_step(batch)
torch.cuda.synchronize()

runs = []
for label, opts in SETTINGS:
    old = {k: L.get_option(k) for k in opts}
    for k, v in opts.items():
        L.set_option(k, v)
    tr.training_step(batch)
    torch.cuda.synchronize()
    acc = None
    REP = 3
    for _ in range(REP):
        PROFILER.start()
        tr.training_step(batch)
        torch.cuda.synchronize()
        PROFILER.stop()
        recs = [(name, fl, by, a.elapsed_time(b)) for name, fl, by, a, b in PROFILER.records]
        if acc is None:
            acc = [[r[0], r[1], r[2], r[3]] for r in recs]
        else:
            assert len(acc) == len(recs)
            for x, r in zip(acc, recs):
                x[3] = min(x[3], r[3])
    for k, v in old.items():
        L.set_option(k, v)
    runs.append(acc)
    print(f"{label:20s} launches {len(acc)} engine ms {sum(x[3] for x in acc):.2f}", flush=True)

n = len(runs[0])
assert all(len(r) == n for r in runs), [len(r) for r in runs]
best = sum(min(r[i][3] for r in runs) for i in range(n))
print(f"per-launch best of all settings: {best:.2f} ms (default {sum(x[3] for x in runs[0]):.2f})")
groups = {}
for i in range(n):
    key = (runs[0][i][0], runs[0][i][1], runs[0][i][2])
    g = groups.setdefault(key, [0] + [0.0] * len(runs) + [set()])
    g[0] += 1
    for j, r in enumerate(runs):
        g[1 + j] += r[i][3]
        if j:
            g[-1].add((j, r[i][0]))
print(f"{'default kernel':36s} {'n':>3s} {'GFLOP':>7s} {'MB':>6s} | us per launch: " + " | ".join(s[0] for s in SETTINGS))
for key, g in sorted(groups.items(), key=lambda kv: -(kv[1][1] - min(kv[1][1:-1])))[:50]:
    name, fl, by = key
    per = [1e3 * t / g[0] for t in g[1:-1]]
    print(f"{name:36s} {g[0]:3d} {fl / 1e9:7.3f} {by / 1e6:6.1f} | " + " ".join(f"{t:7.1f}" for t in per) + f" | save {g[1] - min(g[1:-1]):.2f} ms")
