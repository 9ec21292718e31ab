#!/usr/bin/env python3
"""Debug builds: link a variant of libvisinger_hip.so in which some translation units are compiled with extra -D flags (timing-only
perturbations, phase stamps) into build/<name>/libvisinger_hip.so; load it with VS_LIB=build/<name>/libvisinger_hip.so.
Usage: python tools/build_variant.py <name> -DFLAG[=v] ... unit.hip [unit2.hip ...]      (the production library is untouched)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visinger_amd.csrc import build as B  # noqa: E402

name = sys.argv[1]
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
units = [a for a in sys.argv[2:] if a.endswith(".hip")]
B.build(verbose=False)
out = os.path.join(ROOT, "build", name)
os.makedirs(out, exist_ok=True)
objs, procs = [], []
for src in B.sources():
    base = os.path.basename(src)
    if base in units:
        obj = os.path.join(out, base[:-4] + ".o")
        procs.append(subprocess.Popen([B.HIPCC] + [f for f in B.FLAGS if f != "-shared"] + defs + ["-c", src, "-o", obj]))
    else:
        obj = src[:-4] + ".o"
    objs.append(obj)
for pr in procs:
    if pr.wait() != 0:
        sys.exit("hipcc failed")
lib = os.path.join(out, "libvisinger_hip.so")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [os.path.join(B.HERE, "build_stamp.gen.o")])
print(lib)
