#!/usr/bin/env python3
"""Where does hipcc wait for ALL outstanding vector-memory operations inside a loop?  Compiles one translation unit of visinger_amd/csrc to gfx950
assembly and lists, per kernel, the `s_waitcnt vmcnt(0)` instructions that sit in blocks marked `in Loop` -- the pattern behind every entry of
DESIGN.md 4.4 (a load next to stores, or behind a branch, waits for everything in flight: one memory round trip per iteration).  A report, not a gate:
epilogue loops legitimately end in such waits; read the ISA around the hits (hipcc -S output is kept in --out).
Usage: python tools/isa_lint.py conv_backward.hip [--out /tmp/isa] [--top 20]"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visinger_amd.csrc import build as B  # noqa: E402


def scan(path):
    name, inloop, hits = None, False, {}
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name = m.group(1)
            continue
        if line.startswith(".LBB"):
            inloop = ("in Loop" in line) or ("Loop Header" in line)
            continue
        if name and inloop and "s_waitcnt" in line and "vmcnt(0)" in line:
            hits.setdefault(name, []).append(ln)
    return hits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("unit")
    ap.add_argument("--out", default="/tmp/isa_lint")
    ap.add_argument("--top", type=int, default=20)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    src = os.path.join(B.HERE, a.unit)
    asm = os.path.join(a.out, a.unit[:-4] + ".s")
    subprocess.check_call([B.HIPCC] + [f for f in B.FLAGS if f != "-shared"] + ["-S", "--cuda-device-only", src, "-o", asm], stderr=subprocess.DEVNULL)
    hits = scan(asm)
    for k, v in sorted(hits.items(), key=lambda kv: -len(kv[1]))[:a.top]:
        dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{len(v):5d}  {dem[:120]}   (first at {asm}:{v[0]})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
