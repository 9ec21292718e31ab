#!/usr/bin/env python3
"""Where the synthesis step goes, by sub-module (HIP events; B=32, T_mel=1024, hop 256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
model, hp = bench.build_model(); model = model.cuda()
B, T = 32, 1024
text, pitch, dur, mel2ph, spk, noise = bench.synthetic_batch(B, T, T // 8, 64, 1234, "cuda")
times = {}
def timed(name, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
    times[name] = times.get(name, 0.0) + e0.elapsed_time(e1)
    return out
def step():
    with torch.no_grad():
        m = model
        nonpad = (mel2ph > 0).float().unsqueeze(1)
        prior = timed("text_encoder", lambda: m.text_encoder(text, pitch, dur, mel2ph)) * nonpad
        pos = m.embed_positions(prior.shape[0], prior.shape[2], prior.transpose(1, 2)[..., 0])
        prior = prior + pos.transpose(1, 2)
        spk_emb = m.speaker_embedding(None, spk).transpose(1, 2)
        ret = {}
        cond = timed("pitch_predictor", lambda: m.forward_pitch(prior, None, None, spk_emb, nonpad, ret)).transpose(1, 2)
        mu, logs = timed("frame_prior", lambda: m.frame_prior(prior, nonpad, cond))
        z_p = (mu + noise * torch.exp(logs)) * nonpad
        z_q = timed("flow_inverse", lambda: m.flow(z_p, nonpad, g=spk_emb, reverse=True)) * nonpad
        dec = m.decoder
        # generator by stage (decoder.py:40-59)
        from visinger_amd.modules.visinger import decoder as D
        return timed("generator", lambda: dec(z_q * nonpad, g=spk_emb))
for _ in range(2): step()
times.clear()
N = 3
for _ in range(N): step()
tot = sum(times.values())
for k, v in times.items(): print(f"{k:16s} {v / N:8.2f} ms  {100 * v / tot:5.1f} %")
print(f"{'sum':16s} {tot / N:8.2f} ms")
