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


# conv_split_kernel issues its weight-fragment loads as inline asm and waits for them with hand-counted `s_waitcnt vmcnt(n)`:
# a register spill inside its main loop would be a vector-memory instruction hipcc adds behind the count's back.  The build
# fails instead of shipping such a kernel.
NO_SCRATCH = {"conv_split.hip": "conv_split_kernel", "conv_wsplit.hip": "conv_wsplit_kernel",
              # (no hand-counted waits here, but a spill in this kernel is paid once per conv of a fused chain: round 3 found 112 B / lane
              #  = 0.5 GB of scratch stores per launch behind a run-time LDS pitch)
              "resblock_f16.hip": "resblock_"}                # resblock_f16_kernel and resblock_bf16_kernel


def check_no_scratch(src, remarks):
    name, bad = None, []
    for line in remarks.splitlines():
        if "remark:" not in line:
            if "warning" in line or "error" in line:
                sys.stderr.write(line + "\n")
            continue
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize [bytes/lane]:" in line and name and NO_SCRATCH[os.path.basename(src)] in name:
            if int(line.split("ScratchSize [bytes/lane]:")[1].split()[0]) != 0:
                bad.append(name)
    if bad:
        raise RuntimeError(f"{src}: kernels with hand-counted vmcnt waits must not use scratch: {bad}")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.inc")) + glob.glob(os.path.join(HERE, "..", "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def includes(path, seen=None):
    """the local files `path` includes, transitively (quoted includes and the VS_EPILOGUE_INC-style macro definitions)"""
    import re
    seen = set() if seen is None else seen
    try:
        text = open(path).read()
    except OSError:
        return seen
    for name in re.findall(r'#\s*(?:include|define\s+\w+_INC)\s+"([^"]+)"', text):
        dep = os.path.normpath(os.path.join(os.path.dirname(path), name))
        if dep not in seen and os.path.exists(dep):
            seen.add(dep)
            includes(dep, seen)
    return seen


def stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src, os.path.abspath(__file__)] + sorted(includes(src)))


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if not force and not stale(src, obj):      # (objects are per translation unit: only the units whose sources changed)
            continue
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
        guarded = os.path.basename(src) in NO_SCRATCH
        if guarded:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stderr=subprocess.PIPE if guarded else None, text=True), guarded))
    for src, pr, guarded in procs:
        err = pr.communicate()[1] if guarded else None
        if pr.wait() != 0:
            if err:
                sys.stderr.write(err)
            raise RuntimeError(f"hipcc failed on {src}")
        if guarded:
            check_no_scratch(src, err)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.utime(LIB, None)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
