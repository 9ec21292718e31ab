import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
model, hp = bench.build_model(); model = model.cuda()
B, T = int(os.environ.get("GB", 1)), int(os.environ.get("GT", 1024))
text, pitch, dur, mel2ph, spk, noise = bench.synthetic_batch(B, T, T // 8, 64, 1234, "cuda")
def step():
    with torch.no_grad():
        return model(text, pitch, dur, mel2ph, spk_id=spk, infer=True, noise=noise)["wav_out"]
for _ in range(3): ref = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 10
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph vs eager max diff", float((out - ref).abs().max()))
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); graphed = (time.perf_counter() - t0) / 10
print(f"B={B} T={T}: eager {eager*1e3:.2f} ms, graphed {graphed*1e3:.2f} ms")
