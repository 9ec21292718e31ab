#!/usr/bin/env python3
"""Where the weight-gradient time of the GAN training step goes, by conv shape (HIP events around every vs_conv_wgrad call of one
step; BASELINE config 3 sizes)."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import autograd as ag, ops
from visinger_amd.models.visinger import hop256_hparams
from visinger_amd.train import VISingerTrainer, synthetic_train_batch

B, T = 16, 512
hp = hop256_hparams(p_dropout=0.0)
torch.manual_seed(1234)
tr = VISingerTrainer(64, 117, 131, hp).cuda().configure().train()
batch = synthetic_train_batch(B, T, T // 8, tr.hop, 64, hp["num_linear_bins"], 1234, "cuda")
for _ in range(2):
    tr.training_step(batch)
rec = []
orig = ops.conv_wgrad


def timed(gy, x, k, dil=1, pad=0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(gy, x, k, dil, pad)
    e1.record()
    rec.append(((tuple(gy.shape), tuple(x.shape), k, dil), e0, e1))
    return out


ag.conv_wgrad = timed
tr.training_step(batch)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    agg[key][0] += 1
    agg[key][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"{len(rec)} wgrad calls, {tot:.1f} ms (incl. the plane sums)")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    (gys, xs, k, d) = key
    fl = 2.0 * gys[0] * gys[1] * xs[1] * k * gys[2]
    print(f"  gy{gys} x{xs} k{k} d{d}: {n:3d} calls {ms:7.2f} ms  {fl * n / ms / 1e9:6.1f} TFLOP/s")
