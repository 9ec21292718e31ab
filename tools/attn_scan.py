#!/usr/bin/env python3
"""Fixed and per-tile cost of the split attention kernels: launches of 512 workgroups (2 heads x 96 channels, B * T / 128 = 256 query blocks) at
T = 256 .. 4096, i.e. 8 .. 128 key tiles per workgroup.  Run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3
tools/attn_scan.py`; `python3 tools/attn_scan.py --read DIR` then prints the median duration per shape and the two-point fit."""
import os, sys, glob, csv, statistics
N = 8
SHAPES = [(128, 256), (64, 512), (32, 1024), (16, 2048), (8, 4096)]
WIDE = "--wide" in sys.argv      # BASELINE configs[4]'s heads (2 x 256 channels) in the plain-bf16 arithmetic: relattn_dma_kernel<8, 2>
if WIDE:
    SHAPES = [(32, 1024), (16, 2048), (8, 4096)]
if "--read" in sys.argv:
    d = sys.argv[sys.argv.index("--read") + 1]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "relattn" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
    i = 0
    for math in (("plain bf16",) if WIDE else ("split-bf16 x6", "split-f16 x3")):
        pts = []
        for B, T in SHAPES:
            chunk = dur[i:i + N + 2][2:]
            i += N + 2
            med = statistics.median(x[1] for x in chunk)
            pts.append((T // 32, med))
            print(f"{math:14s} {chunk[0][0][10:45]:36s} B={B:4d} T={T:5d} tiles/WG={T // 32:4d}: median {med:8.1f} us  min {min(x[1] for x in chunk):8.1f}")
        b = (pts[-1][1] - pts[0][1]) / (pts[-1][0] - pts[0][0])
        print(f"  -> per tile {b:.2f} us, fixed {pts[0][1] - b * pts[0][0]:.1f} us")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import rel_attention
nh, C = (2, 512) if WIDE else (2, 192)
for math in ((L.MATH_BF16,) if WIDE else (L.MATH_SPLIT6, L.MATH_SPLIT3)):
    for B, T in SHAPES:
        qkv = torch.randn(B, 3 * C, T, device="cuda")
        rk = torch.randn(1, 9, C // nh, device="cuda") * 0.1
        rv = torch.randn(1, 9, C // nh, device="cuda") * 0.1
        mask = torch.ones(B, T, device="cuda")
        out = torch.empty(B, C, T, device="cuda")
        for _ in range(N + 2):
            rel_attention(qkv, nh, rk, rv, mask, 4, out=out, math=math)
        torch.cuda.synchronize()
