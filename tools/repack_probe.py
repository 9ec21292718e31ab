#!/usr/bin/env python3
"""Which conv handles re-pack their weights in a steady-state synthesis step (they should not: the parameters do not change)?  Wraps
ConvOp.set_weights after two warm-up steps and prints every call that misses the handle's cache key.  Usage (GPU box): python tools/repack_probe.py [2]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from visinger_amd import ops
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
wl = bench.InferenceWorkload(cfg, 8 if cfg == 2 else 32, 512 if cfg == 2 else 1024, 192, "split3", "f32", 256, False, dev)
for _ in range(2): wl.step()
torch.cuda.synchronize()
seen = collections.Counter()
orig = ops.ConvOp.set_weights
def set_weights(self, w, g=None, bias=None, force=False):
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in (w, g, bias))
    if key != self._wkey or force:
        seen[(self.kind, self.c_in, self.c_out, self.k, self.dil, self.flags, bool(force), self._wkey is None)] += 1
    return orig(self, w, g, bias, force)
ops.ConvOp.set_weights = set_weights
for _ in range(3): wl.step()
torch.cuda.synchronize()
print("re-packs in 3 steady-state steps (kind, c_in, c_out, k, dil, flags, force, key was None): count")
for k, v in seen.items(): print(" ", k, v)
print("total", sum(seen.values()))
