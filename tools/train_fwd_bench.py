import sys, os, time, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/visinger_amd") else os.getcwd())
import bench
from visinger_amd.ops import PROFILER
model, hp = bench.build_model(); model = model.cuda()
B, T = 16, 512
text, pitch, dur, mel2ph, spk, noise = bench.synthetic_batch(B, T, T // 8, 64, 1234, "cuda")
lin = torch.randn(B, T, 1025, device="cuda").abs()
def step():
    with torch.no_grad():
        return model(text, pitch, dur, mel2ph, spk_id=spk, mel=lin, infer=False)
for _ in range(2): r = step()
torch.cuda.synchronize(); PROFILER.start(); t0 = time.perf_counter()
for _ in range(5): r = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5; PROFILER.stop()
print("training-side forward (posterior + phoneme + flow fwd + prior + segment decode), B=16 T=512: %.2f ms" % (dt * 1e3))
for k, v in PROFILER.summary().items(): print("  ", k, "%.2f ms/step  %.1f TF" % (v["ms"] / 5, v["flops"] / v["ms"] / 1e9))
