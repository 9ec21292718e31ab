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
NO_SCRATCH = {"conv_split.hip": "conv_split_kernel",
              # (no hand-counted waits here, but a spill in this kernel is paid once per conv of a fused chain: round 3 found 112 B / lane
              #  = 0.5 GB of scratch stores per launch behind a run-time LDS pitch)
              "resblock_f16.hip": "resblock_",                # resblock_f16_kernel and resblock_bf16_kernel
              # (a spill inside the chunk body would sit between MFMAs that leave it no issue slot; ADVICE r4: the LDS-DMA ring of the attention
              #  kernel counts its vector-memory operations by hand as well)
              "conv_ktap.hip": "conv_ktap_kernel", "conv_ktap_bf16.hip": "conv_ktap_kernel", "conv_ktap_small.hip": "conv_ktap_kernel", "conv_ktap_pair.hip": "conv_ktap_kernel", "attention_dma.hip": "relattn_dma_kernel"}


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


# gfx950, round 6 (tools/ubench/pk_opsel_probe.hip, conv_common.h): v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 with the op_sel bit of their SECOND source set (the low
# result reads the high register of src1) return wrong low results in lanes 48-63 beside a wave that issues MFMAs.  hipcc's SLP vectoriser forms them from scalar code
# whenever it likes, so every object is disassembled and the build fails on one.
LLVM_BIN = os.environ.get("VS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PK_OPSEL_SRC1 = r"\bv_pk_(?:mul|add|fma)_f32\b[^\n]*\bop_sel:\[[01],1"


def device_isa(obj):
    """gfx950 disassembly of the device code bundled in a host object"""
    fat, co = obj + ".fat.tmp", obj + ".co.tmp"
    try:
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + fat, "--output=" + co], stderr=subprocess.DEVNULL)
        return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
    finally:
        for f in (fat, co):
            if os.path.exists(f):
                os.remove(f)


def packed_opsel_hits(isa):
    """(kernel symbol, instruction) of every packed-fp32 instruction whose low result reads the high register of src1"""
    import re
    hits, name = [], None
    for line in isa.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
        elif re.search(PK_OPSEL_SRC1, line):
            hits.append((name, line.split("//")[0].strip()))
    return hits


def check_packed_opsel(src, obj):
    hits = packed_opsel_hits(device_isa(obj))
    if hits:
        kernels = sorted({k for k, _ in hits})
        raise RuntimeError(f"{src}: {len(hits)} packed-fp32 instruction(s) with op_sel:[_,1] (wrong low results in lanes 48-63 beside MFMAs on gfx950; "
                           f"conv_common.h: mul_f32_scalar / add_f32_scalar) in {len(kernels)} kernel(s), e.g. {kernels[0]}: {hits[0][1]}")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def all_inputs():
    """every file the library is compiled from (kernels, shared headers / textual includes, the public header, this recipe)"""
    return sorted(sources() + glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.inc")) +
                  glob.glob(os.path.join(HERE, "..", "..", "include", "*.h")) + [os.path.abspath(__file__)])


def source_hash():
    """sha256 over the names and contents of all_inputs(): compiled into the library (vs_source_hash()) and compared by
    visinger_amd._lib.lib() at load -- a library built from other sources (a stale object that survived a checkout with a fresh mtime)
    is refused instead of run.  Content-based: independent of mtimes and of where the tree sits."""
    import hashlib
    h = hashlib.sha256()
    for path in all_inputs():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


STAMP = os.path.join(HERE, ".build_stamp.json")      # {"source_hash": ..., "objects": {src basename: hash of its own inputs}}


def _read_stamp():
    import json
    try:
        with open(STAMP) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def needs_build():
    return not os.path.exists(LIB) or _read_stamp().get("source_hash") != source_hash()


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


def unit_hash(src):
    """hash of one translation unit's inputs (its source, everything it includes, the compiler flags)"""
    import hashlib
    h = hashlib.sha256(" ".join([HIPCC] + FLAGS).encode())
    for path in [src] + sorted(includes(src)):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def build(force=False, verbose=True):
    import json
    if not force and not needs_build():
        return LIB
    stamp = _read_stamp().get("objects", {}) if not force else {}
    # the hash the library will answer with is taken BEFORE anything is compiled: a source edited while the compilers run (round 5: it happened) then leaves a
    # library whose hash differs from the tree's -- refused at load, rebuilt by the next call -- instead of a stale object under a fresh-looking stamp
    digest = source_hash()
    objs, procs, hashes = [], [], {}
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        hashes[os.path.basename(src)] = unit_hash(src)
        if os.path.exists(obj) and stamp.get(os.path.basename(src)) == hashes[os.path.basename(src)]:
            continue      # (objects are per translation unit: only the units whose inputs changed, by content)
        # compile to a temporary name: the object only takes its final name once hipcc AND the scratch guard have passed, so a failed
        # build can never leave a fresh-looking object of a spilling kernel behind for the next incremental build to link
        tmp = obj + ".tmp"
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", tmp]
        guarded = os.path.basename(src) in NO_SCRATCH
        if guarded:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, tmp, subprocess.Popen(cmd, stderr=subprocess.PIPE if guarded else None, text=True), guarded))
    failed = None
    for src, obj, tmp, pr, guarded in procs:
        err = pr.communicate()[1] if guarded else None
        try:
            if pr.wait() != 0:
                if err:
                    sys.stderr.write(err)
                raise RuntimeError(f"hipcc failed on {src}")
            if guarded:
                check_no_scratch(src, err)
            check_packed_opsel(src, tmp)
            os.replace(tmp, obj)
        except RuntimeError as e:
            failed = failed or e
            for stale_file in (tmp, obj):
                if os.path.exists(stale_file):
                    os.remove(stale_file)
    if failed:
        if os.path.exists(STAMP):
            os.remove(STAMP)
        raise failed
    # the source hash the library answers with (vs_source_hash): a generated C file, compiled by the host compiler only
    stamp_c = os.path.join(HERE, "build_stamp.gen.c")
    with open(stamp_c, "w") as f:
        f.write('/* generated by build.py */\n__attribute__((visibility("default"))) const char *vs_source_hash(void) { return "%s"; }\n' % digest)
    stamp_o = os.path.join(HERE, "build_stamp.gen.o")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-c", stamp_c, "-o", stamp_o])
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs + [stamp_o]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as f:
        json.dump({"source_hash": digest, "objects": hashes}, f, indent=1)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
