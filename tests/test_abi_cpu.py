"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/*.h declares
(no compute without a GPU), the product path fails loudly without a GPU, and the nn.Module mirrors expose exactly
the reference's parameter names/shapes (manifest generated from the reference by tests/golden/make_golden.py)."""
import ctypes
import inspect
import json
import os
import re

import pytest
import torch

from conftest import GOLDEN, ROOT


def declared_symbols():
    syms = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            syms |= set(re.findall(r"VS_API\s+[\w\s\*]+?\b(vs_\w+)\s*\(", txt))
    return syms


def test_library_exports_every_declared_symbol():
    from visinger_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in sorted(syms) if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    assert _lib.lib().vs_abi_version() == _lib.EXPECTED_ABI == 7


def test_library_carries_the_hash_of_its_sources_and_a_foreign_build_is_refused(monkeypatch):
    """binaries built in the container travel to the GPU box; an mtime check passes a stale object after a checkout.
    The library answers with the sha256 of what it was compiled from and the loader compares it with the tree."""
    from visinger_amd import _lib
    from visinger_amd.csrc import build
    L = _lib.lib()
    assert L.vs_source_hash().decode() == build.source_hash() and len(build.source_hash()) == 64
    assert not build.needs_build()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 64)       # "the tree changed under the library"
    with pytest.raises(_lib.VisingerHipError, match="built from other sources"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib().vs_abi_version() == _lib.EXPECTED_ABI


def test_a_failed_scratch_guard_leaves_no_object_behind(tmp_path, monkeypatch):
    """hipcc writes the object before the NO_SCRATCH guard can fail; the next incremental build must not find it.
    A fake compiler that 'succeeds' with a spilling kernel in its remarks: the build raises and neither the .o nor its .tmp exists."""
    import stat
    from visinger_amd.csrc import build
    src = tmp_path / "conv_split.hip"
    src.write_text("// fake\n")
    fake = tmp_path / "fakecc"
    fake.write_text("#!/bin/sh\nwhile [ $# -gt 0 ]; do if [ \"$1\" = -o ]; then out=$2; fi; shift; done\necho obj > $out\n"
                    "echo 'x.hip:1:1: remark: Function Name: _ZN2vs17conv_split_kernelILi1EEEvv [-Rpass-analysis=kernel-resource-usage]' 1>&2\n"
                    "echo 'x.hip:1:1: remark:     ScratchSize [bytes/lane]: 112 [-Rpass-analysis=kernel-resource-usage]' 1>&2\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr(build, "HERE", str(tmp_path))
    monkeypatch.setattr(build, "LIB", str(tmp_path / "lib.so"))
    monkeypatch.setattr(build, "STAMP", str(tmp_path / ".build_stamp.json"))
    monkeypatch.setattr(build, "HIPCC", str(fake))
    with pytest.raises(RuntimeError, match="must not use scratch"):
        build.build(verbose=False)
    assert not (tmp_path / "conv_split.o").exists() and not (tmp_path / "conv_split.o.tmp").exists() and not (tmp_path / "lib.so").exists()


def test_arguments_are_validated_without_a_gpu():
    from visinger_amd import _lib
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.vs_conv_create(ctypes.byref(h), 0, 16, 16, 33, 5, 80, 0) == 3          # VS_EUNSUPPORTED (span)
    assert b"span" in L.vs_last_error()
    assert L.vs_conv_create(ctypes.byref(h), 7, 16, 16, 3, 1, 1, 0) == 1            # VS_EINVAL (kind)
    assert L.vs_conv_create(ctypes.byref(h), 2, 16, 15, 3, 1, 1, 0) == 1            # PAIRED needs even c_out
    assert L.vs_conv_create(ctypes.byref(h), 0, 16, 16, 3, 1, 1, 0) == 0
    assert L.vs_conv_out_len(h, 100) == 100
    L.vs_conv_destroy(h)
    assert L.vs_conv_create(ctypes.byref(h), 1, 16, 8, 16, 8, 4, 0) == 0
    assert L.vs_conv_out_len(h, 10) == 80
    L.vs_conv_destroy(h)
    # vs_conv_set_weights_pair: the second handle must be the VS_CONV_ADJOINT counterpart of the first (a mismatched pair made the
    # pack read the weight out of bounds) -- rejected before anything is reserved or launched
    h0, h1, h2 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.vs_conv_create(ctypes.byref(h0), 0, 32, 64, 3, 1, 1, 0) == 0
    assert L.vs_conv_create(ctypes.byref(h1), 0, 64, 32, 3, 1, 1, 0) == 0            # swapped channels, but not an ADJOINT handle
    assert L.vs_conv_create(ctypes.byref(h2), 0, 64, 48, 3, 1, 1, 4) == 0            # ADJOINT (flag 4), wrong c_out
    w = (ctypes.c_float * (64 * 32 * 3))()
    fp = ctypes.cast(w, ctypes.POINTER(ctypes.c_float))
    L.vs_conv_set_weights_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]
    assert L.vs_conv_set_weights_pair(h0, h1, fp, None, None) == 1 and b"ADJOINT" in L.vs_last_error()
    assert L.vs_conv_set_weights_pair(h0, h2, fp, None, None) == 1
    # vs_conv_set_weights_batch: the same handle twice in one batch is refused before anything is reserved or launched (each job flips its handle's
    # weight-maximum slot, a duplicate made one job clear the slot the other reads)
    hs = (ctypes.c_void_p * 3)(h0, h1, h0)
    ws = (ctypes.POINTER(ctypes.c_float) * 3)(fp, fp, fp)
    L.vs_conv_set_weights_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert L.vs_conv_set_weights_batch(hs, ws, None, 3, None) == 1 and b"twice" in L.vs_last_error()
    for hh in (h0, h1, h2):
        L.vs_conv_destroy(hh)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    from visinger_amd import _lib
    from visinger_amd.modules.visinger.encoder import WaveNet
    m = WaveNet(16, 5, 1, 2).eval()
    with torch.no_grad(), pytest.raises(_lib.VisingerHipError):
        m(torch.zeros(1, 16, 8), torch.ones(1, 1, 8))


def test_state_dict_matches_reference_manifest():
    """Same parameter names and shapes as the reference's VISinger -> its checkpoints load with strict=True."""
    from visinger_amd.models.visinger import VISinger
    man = json.load(open(os.path.join(GOLDEN, "visinger_state_dict_manifest.json")))
    m = VISinger(man["ph_dict_size"], man["pitch_size"], man["dur_size"], man["hparams"])
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert ours == man["state_dict"]
    # and the reverse direction: a reference-shaped state dict loads strictly
    m.load_state_dict({k: torch.zeros(s) for k, s in man["state_dict"].items()}, strict=True)


def test_signatures_match_survey_8b():
    """Constructor / forward argument names of the boundary classes."""
    from visinger_amd.modules.visinger import encoder, flow, decoder, predictor
    from visinger_amd.modules import rel_transformer as rt

    def params(f):
        return [p for p in inspect.signature(f).parameters if p != "self"]

    assert params(encoder.WaveNet.__init__) == ["hidden_channels", "kernel_size", "dilation_rate", "n_layers", "gin_channels", "p_dropout"]
    assert params(encoder.WaveNet.forward)[:3] == ["x", "x_mask", "g"]
    assert params(encoder.PosteriorEncoder.__init__) == ["in_channels", "out_channels", "hidden_channels", "kernel_size", "dilation_rate", "n_layers", "gin_channels"]
    assert params(encoder.PosteriorEncoder.forward)[:3] == ["x", "nonpadding", "g"]
    assert params(flow.ResidualCouplingBlock.__init__) == ["channels", "hidden_channels", "kernel_size", "dilation_rate", "n_layers", "n_flows", "gin_channels"]
    assert params(flow.ResidualCouplingBlock.forward) == ["x", "x_mask", "g", "reverse"]
    assert params(flow.ResidualCouplingLayer.__init__) == ["channels", "hidden_channels", "kernel_size", "dilation_rate", "n_layers", "p_dropout", "gin_channels", "mean_only"]
    assert params(decoder.Generator.__init__) == ["initial_channel", "resblock", "resblock_kernel_sizes", "resblock_dilation_sizes", "upsample_rates", "upsample_initial_channel", "upsample_kernel_sizes", "gin_channels"]
    assert params(decoder.Generator.forward)[:2] == ["x", "g"]     # (+ optional x_mask: exact decode of a PADDED batch, synth.py)
    assert params(rt.RelativeEncoder.__init__)[:10] == ["hidden_channels", "filter_channels", "n_heads", "n_layers", "kernel_size", "p_dropout", "window_size", "block_length", "pre_ln", "gin_channels"]
    assert params(rt.RelativeEncoder.forward) == ["x", "x_mask", "g"]
    assert params(rt.MultiHeadAttention.__init__) == ["channels", "out_channels", "n_heads", "window_size", "heads_share", "p_dropout", "block_length", "proximal_bias", "proximal_init"]
    assert params(rt.MultiHeadAttention.forward)[:3] == ["x", "c", "attn_mask"]
    assert params(encoder.FramePriorNetwork.__init__) == ["hidden_channels", "filter_channels", "n_heads", "n_layers", "kernel_size", "gin_channels", "p_dropout"]
    assert params(encoder.TextEncoder.forward) == ["text_tokens", "pitch_tokens", "dur_tokens", "mel2ph"]
    assert params(predictor.PitchPredictor.forward) == ["x", "x_mask", "spk_emb"]
    assert params(predictor.PhonemePredictor.forward) == ["x", "x_mask"]
    with pytest.raises(AssertionError):
        flow.ResidualCouplingLayer(5, 8, 5, 1, 2)                 # channels % 2 (flow.py:50)
    with pytest.raises(AssertionError):
        encoder.WaveNet(8, 4, 1, 2)                               # even kernel (encoder.py:133)
    with pytest.raises(AssertionError):
        rt.MultiHeadAttention(10, 10, 3)                          # channels % n_heads (rel_transformer.py:107)


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="needs the reference checkout (build container only)")
def test_reference_glue_imports_our_modules_unchanged():
    """Drop-in: the reference's own models/visinger.py, imported unchanged, builds its VISinger out of our classes."""
    import subprocess
    import sys
    code = r'''
import sys
from unittest.mock import MagicMock
sys.dont_write_bytecode = True
for n in ("librosa", "librosa.filters", "pyloudnorm", "webrtcvad", "skimage", "skimage.transform", "parselmouth", "pyworld", "torchaudio"):
    sys.modules.setdefault(n, MagicMock())
sys.path.insert(0, %r)
import visinger_amd
visinger_amd.install_as_reference_modules()
sys.path.append("/root/reference")
import json
from models.visinger import VISinger          # the reference's file
import visinger_amd.modules.visinger.decoder as d
man = json.load(open(%r))
m = VISinger(man["ph_dict_size"], man["pitch_size"], man["dur_size"], man["hparams"])
assert type(m.decoder) is d.Generator, type(m.decoder)
assert {k: list(v.shape) for k, v in m.state_dict().items()} == man["state_dict"]
print("ok")
''' % (ROOT, os.path.join(GOLDEN, "visinger_state_dict_manifest.json"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_conv_handles_survive_copy_and_pickle():
    """modules cache ConvOp handles (forward, backward-data, discriminator); copy.deepcopy (EMA), torch.save(module) and
    pickling for a spawned worker must neither pickle a ctypes pointer nor share a vs_conv_t between two owners."""
    import copy
    import pickle
    import torch
    from visinger_amd import _lib as L
    from visinger_amd.modules.hipconv import HipConv1d, repack_weights
    from visinger_amd.ops import ConvOp
    op = ConvOp(L.CONV1D, 8, 16, 3, 1, 1)
    op._pending_math = L.MATH_F32
    for clone in (copy.deepcopy(op), pickle.loads(pickle.dumps(op))):
        assert clone is not op and clone._args() == op._args() and clone._h is None and clone._pending_math == L.MATH_F32
        assert clone._wkey is None                                   # a copy re-binds its owner's parameters on first use
    m = HipConv1d(8, 16, 3, padding=1)
    m.__dict__["_hip_ops"] = {(0, 0): op}
    m.__dict__["_hip_bwd_ops"] = {"dx": ConvOp(L.CONV1D, 16, 8, 3, 1, 1)}
    m2 = pickle.loads(pickle.dumps(m))
    assert not any(k in m2.__dict__ for k in ("_hip_ops", "_hip_bwd_ops", "_hip_disc_ops"))
    assert torch.equal(m2.weight, m.weight)
    plain = torch.nn.Conv1d(4, 4, 3)                                  # the discriminators' holders are stock nn.Conv1d / nn.Conv2d
    plain.__dict__["_hip_disc_ops"] = {"fwd": op}
    p2 = copy.deepcopy(plain)
    assert p2.__dict__["_hip_disc_ops"]["fwd"] is not op and p2.__dict__["_hip_disc_ops"]["fwd"]._h is None
    op._wkey = ("stale",)
    repack_weights(m)
    assert op._wkey is None
    # the fused q | k | v projection of an attention layer is keyed on (data_ptr, _version) of six parameters, which a
    # `.data` edit does not change: repack_weights must drop it (and its training twin), and it must never be pickled
    from visinger_amd.modules.rel_transformer import MultiHeadAttention, RelativeEncoder
    enc = RelativeEncoder(16, 32, 2, 1, kernel_size=3)
    att = enc.attn_layers[0]
    att.__dict__["_hip_qkv_inf"] = (("key",), op, torch.zeros(1), torch.zeros(1))
    att.__dict__["_hip_qkv"] = object()
    att.conv_q.weight.data.mul_(2.0)                                 # the edit the cache key cannot see
    a2 = pickle.loads(pickle.dumps(att))
    assert isinstance(a2, MultiHeadAttention) and not any(k in a2.__dict__ for k in ("_hip_qkv_inf", "_hip_qkv"))
    assert torch.equal(a2.conv_q.weight, att.conv_q.weight)
    repack_weights(enc)
    assert not any(k in att.__dict__ for k in ("_hip_qkv_inf", "_hip_qkv"))


def test_dispatch_switches_are_read_once_and_set_through_the_abi():
    """VERDICT r2 hygiene: no getenv / os.environ on a launch path -- the library reads its switches from the environment when it
    is loaded (a child process shows it) and changes them only through vs_set_option; the Python layer likewise."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from visinger_amd import _lib as L\n"
            "a = L.get_option('VS_NO_KTAP'), L.get_option('VS_SMALL_GRID_T6'), L.get_option('VS_CONV_MATH'), L.switch('VS_NO_RESPAIR')\n"
            "os.environ['VS_NO_KTAP'] = '0'; os.environ['VS_NO_RESPAIR'] = ''\n"          # too late: read at load / import
            "b = L.get_option('VS_NO_KTAP'), L.switch('VS_NO_RESPAIR')\n"
            "L.set_option('VS_NO_KTAP', 0); L.set_option('VS_NO_RESPAIR', 0)\n"
            "c = L.get_option('VS_NO_KTAP'), L.switch('VS_NO_RESPAIR')\n"
            "with L.options(VS_CONV_MATH=0, VS_NO_TRAIN_ATTN=1):\n"
            "    d = L.get_option('VS_CONV_MATH'), L.switch('VS_NO_TRAIN_ATTN')\n"
            "e = L.get_option('VS_CONV_MATH'), L.switch('VS_NO_TRAIN_ATTN')\n"
            "try:\n    L.set_option('VS_NO_SUCH_SWITCH', 1); f = 'accepted'\nexcept L.VisingerHipError: f = 'refused'\n"
            "print(a, b, c, d, e, f)\n") % ROOT
    env = dict(os.environ, VS_NO_KTAP="1", VS_NO_RESPAIR="1", PYTHONDONTWRITEBYTECODE="1")
    env.pop("VS_CONV_MATH", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip() == "(1, 512, -1, 1) (1, 1) (0, 0) (0, 1) (-1, 0) refused", out.stdout
    # and the sources: no getenv outside the one-time table initialisation, no os.environ switch outside _lib.py
    import glob
    import re
    for path in glob.glob(os.path.join(ROOT, "visinger_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "visinger_amd", "csrc", "*.inc")):
        text = open(path).read()
        assert len(re.findall(r"\bgetenv\(", text)) == (1 if path.endswith("conv_engine.hip") else 0), path
    for path in glob.glob(os.path.join(ROOT, "visinger_amd", "**", "*.py"), recursive=True):
        if not path.endswith(("_lib.py", os.path.join("csrc", "build.py"))):
            assert "os.environ" not in open(path).read(), path


def test_isa_lint_finds_full_waits_inside_loops(tmp_path):
    """tools/isa_lint.py: the scan behind DESIGN.md 4.4 -- `s_waitcnt vmcnt(0)` counts per kernel, only inside blocks hipcc marks as loop bodies."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = tmp_path / "k.s"
    asm.write_text("\n".join([
        "_Z6kernelv:                             ; @_Z6kernelv",
        "\ts_waitcnt vmcnt(0)",                                        # straight-line code: not reported
        ".LBB0_1:                                ; =>This Inner Loop Header: Depth=1",
        "\tglobal_load_dword v1, v[2:3], off",
        "\ts_waitcnt vmcnt(0)",
        "\ts_waitcnt vmcnt(2) lgkmcnt(0)",
        ".LBB0_2:                                ;   in Loop: Header=BB0_1 Depth=1",
        "\ts_waitcnt vmcnt(0) lgkmcnt(1)",
        ".LBB0_3:",
        "\ts_waitcnt vmcnt(0)",
        "_Z5otherv:",
        ".LBB1_1:                                ;   in Loop: Header=BB1_1 Depth=1",
        "\ts_waitcnt lgkmcnt(0)",
    ]) + "\n")
    assert mod.scan(str(asm)) == {"_Z6kernelv": [5, 8]}


def test_build_gate_refuses_packed_fp32_op_sel_on_the_second_source(tmp_path):
    """csrc/build.py disassembles every object and fails on v_pk_mul/add/fma_f32 ... op_sel:[_,1] (gfx950: wrong low results in lanes 48-63 beside MFMAs;
    tools/ubench/pk_opsel_probe.hip).  The scanner on sample lines, on a freshly compiled object that holds the form, and on the objects of the
    library as built."""
    import subprocess
    from visinger_amd.csrc import build
    isa = "\n".join([
        "0000000000001000 <_Z1kv>:",
        "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]            // 000000001000: D3B14000 18020902",
        "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]                         // broadcast of the low half: fine",
        "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]                            // first source: fine",
        "\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]                            // not arithmetic: fine",
        "0000000000002000 <_Z1gv>:",
        "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1]",
        "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]                  // third source: fine",
        "\tv_pk_add_f32 v[0:1], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]",
    ])
    assert [k for k, _ in build.packed_opsel_hits(isa)] == ["_Z1kv", "_Z1gv", "_Z1gv"]
    src = tmp_path / "k.hip"
    src.write_text('#include <hip/hip_runtime.h>\ntypedef unsigned long long u64;\n__global__ void k(const u64 *a, u64 *y) {\n'
                   '    u64 r; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(r) : "v"(a[threadIdx.x]), "v"(a[threadIdx.x + 64]));\n'
                   '    y[threadIdx.x] = r;\n}\n')
    obj = tmp_path / "k.o"
    subprocess.check_call([build.HIPCC, "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", str(obj)], stderr=subprocess.DEVNULL)
    with pytest.raises(RuntimeError, match=r"op_sel:\[_,1\]"):
        build.check_packed_opsel(str(src), str(obj))
    for unit in ("conv_split", "conv_backward", "resblock_f16"):          # (the units that held such instructions before round 6, and the one that packs by hand)
        assert build.packed_opsel_hits(build.device_isa(os.path.join(build.HERE, unit + ".o"))) == [], unit


def test_weight_range_check_flags_convs_one_scale_cannot_carry():
    """hipconv.select_math_by_weight_range (the load-time check for real checkpoints) on the host: the drop of the weakest output row below the
    conv's largest weight, for plain and weight-normed Conv1d and for ConvTranspose1d (whose output channel is dim 1), all-zero rows ignored; convs beyond the
    bound move to the exact bf16 x3 split, convs already on another arithmetic are left alone."""
    import torch
    from torch.nn.utils import weight_norm
    from visinger_amd import _lib as L
    from visinger_amd.modules.hipconv import HipConv1d, HipConvTranspose1d, select_math_by_weight_range, weight_row_drop_bits
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({
        "plain": HipConv1d(16, 8, 3), "ranged": HipConv1d(16, 8, 3), "normed": weight_norm(HipConv1d(16, 8, 3)), "tr": weight_norm(HipConvTranspose1d(8, 6, 4, 2)),
        "zero_row": HipConv1d(16, 8, 1), "other_math": HipConv1d(16, 8, 3)})
    with torch.no_grad():
        net["ranged"].weight[3] *= 2.0 ** -14                       # one output row 14 bits down
        net["normed"].weight_g[5] *= 2.0 ** -12                     # a weight-norm gain does the same
        net["tr"].weight_v[:, 2] *= 2.0 ** -11                      # ConvTranspose1d: output channel 2 (dim 1); weight norm runs over dim 0
        net["zero_row"].weight[1] = 0.0                             # a pruned row is not a range problem
        net["other_math"].weight[0] *= 2.0 ** -20
    net["other_math"].__dict__["_hip_math"] = L.MATH_F32
    drops = {k: weight_row_drop_bits(m) for k, m in net.items()}
    assert drops["plain"] < 2 and 13 < drops["ranged"] < 16 and 11 < drops["normed"] < 14 and 10 < drops["tr"] < 13 and drops["zero_row"] < 2
    switched = dict(select_math_by_weight_range(net))
    assert set(switched) == {"ranged", "normed", "tr"}
    assert all(net[k].__dict__["_hip_math"] == L.MATH_SPLIT6 for k in switched)
    assert "_hip_math" not in net["plain"].__dict__ and net["other_math"].__dict__["_hip_math"] == L.MATH_F32
    assert select_math_by_weight_range(net) == []                    # idempotent: what it moved is no longer on the split-f16 arithmetic


def test_load_state_dict_runs_the_weight_range_check_by_itself():
    """a checkpoint whose rows one scale per conv cannot carry switches THAT conv to the exact bf16 x3 split when it is loaded (HipConv*._load_from_state_dict ->
    hipconv.auto_select_math); a later, ordinary checkpoint switches it back; an arithmetic the user chose explicitly is never touched"""
    import torch
    from torch.nn.utils import weight_norm
    from visinger_amd import _lib as L
    from visinger_amd.modules.hipconv import HipConv1d, set_conv_math
    torch.manual_seed(1)
    net = torch.nn.Sequential(HipConv1d(16, 8, 3), weight_norm(HipConv1d(8, 8, 5)), HipConv1d(8, 4, 1))
    ordinary = {k: v.clone() for k, v in net.state_dict().items()}
    wide = {k: v.clone() for k, v in ordinary.items()}
    wide["0.weight"][2] *= 2.0 ** -15
    wide["1.weight_g"][4] *= 2.0 ** -13
    net.load_state_dict(wide, strict=True)
    assert net[0].__dict__.get("_hip_math") == L.MATH_SPLIT6 and net[1].__dict__.get("_hip_math") == L.MATH_SPLIT6 and "_hip_math" not in net[2].__dict__
    net.load_state_dict(ordinary, strict=True)
    assert all("_hip_math" not in m.__dict__ for m in net)
    set_conv_math(net, L.MATH_F32)                                       # the user's choice ...
    net.load_state_dict(wide, strict=True)
    assert all(m.__dict__["_hip_math"] == L.MATH_F32 for m in net)       # ... survives a load
    set_conv_math(net, None)
    assert all("_hip_math" not in m.__dict__ and "_hip_math_auto" not in m.__dict__ for m in net)
