"""Build libvisinger_hip.so (gfx950) in-tree with hipcc.  Usage: python -m visinger_amd.csrc.build [--force]"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libvisinger_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.inc")) + glob.glob(os.path.join(HERE, "..", "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.utime(LIB, None)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
