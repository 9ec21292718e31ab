#!/usr/bin/env python3
"""Does the headline batch run faster as two half-batches on two HIP streams than as one batch on one stream?  (Two bench ranks
time-sharing one GPU reach 113 M samples/s where one reaches 106 M: independent launch chains overlap each other's memory and matrix
phases.)  GPU only.   python tools/two_stream_check.py [streams]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model, hp = bench.build_model()
model = model.cuda()
B, T = 32, 1024
batch = [t.cuda() for t in bench.synthetic_batch(B, T, T // 8, 64, 1234, "cpu")]
text, pitch, dur, mel2ph, spk, noise = batch


def run(sl):
    with torch.no_grad():
        return model(text[sl], pitch[sl], dur[sl], mel2ph[sl], spk_id=spk[sl], infer=True, noise=noise[sl])["wav_out"]


def one():
    return run(slice(0, B))


streams = [torch.cuda.Stream() for _ in range(nstreams)]
per = B // nstreams


def many():
    main = torch.cuda.current_stream()
    outs = []
    for i, st in enumerate(streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(run(slice(i * per, (i + 1) * per)))
    for st in streams:
        main.wait_stream(st)
    return outs


def timeit(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = one()
b = torch.cat(many(), 0)
torch.cuda.synchronize()
print("max |one stream - %d streams| = %.3e" % (nstreams, float((a - b).abs().max())))
t1 = timeit(one)
t2 = timeit(many)
t1b = timeit(one)
print(f"one stream, B=32: {t1:.2f} ms ({t1b:.2f} again); {nstreams} streams x B={per}: {t2:.2f} ms  -> {B * T * 256 / t2 / 1e3:.1f} M samples/s against {B * T * 256 / min(t1, t1b) / 1e3:.1f}")
