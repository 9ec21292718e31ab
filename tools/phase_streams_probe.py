#!/usr/bin/env python3
"""Experiment (round 6): the synthesis step split by PHASE over two HIP streams of different priority -- prior transformers + flow of batch i + 1 on a
high-priority stream while the generator of batch i runs on a normal one -- against the batch rotation of visinger_amd.synth.StreamRotation (whole batches
on alternating streams of equal priority).  Usage (GPU box): python tools/phase_streams_probe.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from visinger_amd.synth import StreamRotation
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
wl = bench.InferenceWorkload(0, 32, 1024, 192, "split3", "f32", 256, False, dev)
model = wl.model
for _ in range(3): wl.step()
torch.cuda.synchronize()

def timed(run_all):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run_all(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / steps

def single():
    for _ in range(steps): wl.step()
def rotation():
    rot = StreamRotation(2)
    for _ in range(steps): rot.run(wl.step)
    rot.join()
H, N = torch.cuda.Stream(), torch.cuda.Stream()
orig = model.decoder.forward
def dec(z, g=None, **kw):
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); N.wait_event(ev)
    z.record_stream(N)
    if g is not None: g.record_stream(N)
    with torch.cuda.stream(N):
        return orig(z, g=g, **kw)
def phased():
    model.decoder.forward = dec
    H.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(H):
        for _ in range(steps): wl.step()
    torch.cuda.current_stream().wait_stream(N); torch.cuda.current_stream().wait_stream(H)
    model.decoder.forward = orig
def phased_rot():      # phase split AND two such pairs in rotation
    pass
for rnd in range(4):
    print("round %d: one stream %.2f | batch rotation %.2f | phase streams %.2f ms/step" % (rnd, timed(single), timed(rotation), timed(phased)), flush=True)
