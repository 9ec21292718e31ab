#!/usr/bin/env python3
"""Single-utterance synthesis latency (B=1, T_mel=1024, hop 256: an 11.9 s clip): eager launches vs one replayed HIP graph
(visinger_amd.synth.GraphedStep), and the effect of the short-launch tile choice (VS_NO_SMALL_GRID=1 restores 128-row tiles)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from visinger_amd.synth import GraphedStep

B, T = int(os.environ.get("LB_B", 1)), int(os.environ.get("LB_T", 1024))
model, hp = bench.build_model()
model = model.cuda()
text, pitch, dur, mel2ph, spk, noise = bench.synthetic_batch(B, T, T // 8, 64, 1234, "cuda")
batch = dict(text_tokens=text, pitch_tokens=pitch, dur_tokens=dur, mel2ph=mel2ph, spk_id=spk)


def eager():
    with torch.no_grad():
        return model(text, pitch, dur, mel2ph, spk_id=spk, infer=True, noise=noise)["wav_out"]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


ref = eager()
med, mn = timed(eager)
print(f"B={B} T_mel={T}: eager   median {med:.2f} ms  min {mn:.2f} ms   ({B * T * 256 / 22050:.1f} s of audio)", flush=True)
with torch.no_grad():
    step = GraphedStep(model, batch, noise, False)
out = step(batch, noise)
torch.cuda.synchronize()
assert torch.equal(out, ref)
med, mn = timed(lambda: step(batch, noise))
print(f"B={B} T_mel={T}: graphed median {med:.2f} ms  min {mn:.2f} ms", flush=True)
